"""Batched training driver: the episode loop of the reference's training script (src/main.py:126-311)
over E environments at once, everything resident on the GPU.

  * DQNPolicy      epsilon-greedy stop/go and alternative choice from the Q-values of the consulting agents
                   (src/main.py:164-172, 193-213), batched over every environment;
  * DeviceReplay   replay memory in HBM: observations stay uint8 (3 planes of plane_stride bytes), uniform
                   sampling without replacement on the device (src/agent.py:134-198 keeps a deque of float64
                   arrays on the host);
  * BatchedTrainer one episode = reset + up to max_steps-1 rollout steps; after every step the transitions
                   of the agents that consulted the network are stored (src/main.py:269-299, default mode) and
                   the learner runs once per `update_every` stored transitions (src/agent.py:55-67).

                   Both replay-feeding modes of the script are there: the default one (flm_replay_store) and
                   switch-to-switch (src/main.py:280-296), as well as the crash penalty (src/main.py:271-273;
                   both flm_replay_store_slots): a step's transitions are listed and copied into the ring by
                   kernels, one host read-back (their number) per step.

Differences from the sequential script, all inherent to stepping many environments together: the learn
steps of one rollout step sample a memory that already holds all of that step's transitions (and the
transitions of one step enter the memory grouped by kind, not agent by agent); random numbers come from one
device generator.
"""
import numpy as np
import torch

from . import parallel
from ._lib import call, ptr, query
from .replay import DeviceReplay
from .rollout import MAX_REQ, RolloutBatch


def obs_to_f32(obs_u8, Rp, T1, plane_stride, out=None):
    """uint8 observations [n,3,plane_stride] -> the network's input [n,3*Rp,T1] float32."""
    n = obs_u8.shape[0]
    if out is None:
        out = torch.empty((n, 3 * Rp, T1), dtype=torch.float32, device=obs_u8.device)
    call("flm_obs_to_f32", ptr(obs_u8), n, Rp, T1, plane_stride, ptr(out),
         torch.cuda.current_stream(obs_u8.device).cuda_stream)
    return out


class DQNPolicy:
    """policy(rollout, info) for RolloutBatch.step: one forward pass over the observations of every departing
    agent and of every alternative of every agent in front of a switch, then the epsilon-greedy choice on the
    device (flm_policy_select).  `last` holds what the replay memory needs: a list with one group
    (agents e*A+a, their observations as shown to the network, the actions chosen).  `raw` are views of buffers
    that the next call overwrites."""

    def __init__(self, agent, eps=0.0, seed=0, chunk_bytes=2 << 30):
        if agent.action_space != 2:                        # the rollout's decision is stop (0) or go (1), src/main.py:163-213
            raise ValueError("the rollout's policy chooses between two actions; the agent has %d" % agent.action_space)
        self.agent = agent
        self.eps = float(eps)
        self.chunk_bytes = int(chunk_bytes)
        self.seed = int(seed) & 0xFFFFFFFF
        self._raw = None
        self._last = None
        self._flat = self._sel = self._pool = None
        self._pool_limit = 0

    def q_values(self, ro, obs_u8, out=None, flat=None):
        b = ro.b
        n = obs_u8.shape[0]
        per = 64 * 3 * b.Rp * 4                            # the convolution output dominates the workspace
        chunk = max(1, self.chunk_bytes // per)
        net = self.agent.qnetwork_local
        rails = ro.rails_used()                            # rails beyond these are padding in every observation
        if n <= chunk:
            # (on the stream the rollout's kernels run on: the observations were written there)
            return net.forward_obs(obs_u8, b.plane_stride, rails, out=out, flat=flat, stream=ro.stream)
        q = torch.empty((n, 2), dtype=torch.float32, device=obs_u8.device) if out is None else out[:n]
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            net.forward_obs(obs_u8[lo:hi], b.plane_stride, rails, out=q[lo:hi], flat=flat)
        return q

    def prepare(self, ro):
        """Host work of the coming call that needs none of the step's sizes; RolloutBatch.step calls it before it
        waits for them, while the device is still busy."""
        self._flat = self.agent.qnetwork_local.flat_params()
        if self._sel is None or self._sel[0].numel() < ro.E * ro.A or self._sel[0].device != ro.device:
            self._sel = tuple(torch.empty((ro.E * ro.A,), dtype=torch.int32, device=ro.device) for _ in range(2))

    def _rows_pool(self, ro, rows):
        """Observation rows and Q-values of a step: buffers that only grow.  The observation rows start as zeros, and
        self._pool_limit is how many bytes of every plane have been written since: a step that writes fewer bytes per
        plane than an earlier one (another batch of environments with fewer rails) clears them first."""
        pool = self._pool
        if pool is None or pool[0].shape[0] < rows or pool[0].device != ro.device or pool[0].shape[2] != ro.b.plane_stride:
            self._pool = pool = None                        # (release before taking larger ones)
            cap = rows + rows // 4 + 64
            self._pool = pool = (torch.zeros((cap, 3, ro.b.plane_stride), dtype=torch.uint8, device=ro.device),
                                 torch.empty((cap, 2), dtype=torch.float32, device=ro.device))
            self._pool_limit = 0
        return pool

    def __call__(self, ro, info):
        net, best = info["net"], info["best"]
        req1, req2, alt = info["req_ready"], info["req_switch"], info["alt"]
        n1, n2 = req1.numel(), req2.numel()
        self._raw = self._last = None
        if n1 + n2 == 0:
            return net, best
        if self._flat is None:
            self.prepare(ro)
        flat, self._flat = self._flat, None
        q_done = False
        if alt is not None:
            # departing agents + all alternatives: at most n1 + n2 * cap_alt rows.  The observations are enqueued for that
            # many (the kernel skips the rows beyond the number its predecessor counted), then the host waits for the number
            rows_max = n1 + n2 * ro.cap_alt
            if rows_max <= MAX_REQ:
                obs_pool, q_pool = self._rows_pool(ro, rows_max)
                ro.mark_counts()
                pool = ro._alt_pool                         # (the tables of this step's alternatives, whole buffers)
                # only the rails the environments have are written: the rest of every plane is zero and stays so
                limit = ro.plane_bytes_used()
                if limit < self._pool_limit:
                    obs_pool.zero_()
                self._pool_limit = limit
                ro.consult_observations_counted(pool["row_req"], pool["cols"], pool["row_own"], obs_pool, rows_max, limit)
                # what the forward pass needs apart from the number of rows is put together before the wait, so that
                # the launch follows it at once (the same call as Dueling_DQN.forward_obs makes)
                net_l, b = self.agent.qnetwork_local, ro.b
                fast = rows_max <= max(1, self.chunk_bytes // (64 * 3 * b.Rp * 4))
                if fast:
                    head = (ptr(flat), ptr(obs_pool))
                    tail = (b.Rp, net_l.width, b.plane_stride, net_l.action_space, ro.rails_used(),
                            ptr(net_l._workspace(rows_max)), ptr(q_pool), ro.stream)
                rows = ro.wait_counts()[2]
                if fast and rows:
                    call("flm_dqn_forward_u8_rails", *head, rows, *tail)
                    q_done = True
                obs = obs_pool[:rows]
            else:
                rows = ro.read_counts()[2]
                if rows:
                    obs_pool, q_pool = self._rows_pool(ro, rows)
                    obs = ro.consult_observations(alt["row_req"][:rows], alt["cols"].view(-1, ro.b.T1), out=obs_pool[:rows],
                                                  own_row=alt["row_own"][:rows])
                    self._pool_limit = ro.b.plane_stride
        else:
            rows = n1
            obs_pool, q_pool = self._rows_pool(ro, rows)
            obs = ro.consult_observations(req1, out=obs_pool[:rows])
            self._pool_limit = ro.b.plane_stride            # (whole planes written)
        if rows == 0:                                       # agents before a switch, none with an alternative
            return net, best
        q = q_pool[:rows] if q_done else self.q_values(ro, obs, out=q_pool, flat=flat)
        sel_row, sel_act = self._sel[0][:n1 + n2], self._sel[1][:n1 + n2]
        call("flm_policy_select", ptr(q), ptr(req1) if n1 else None, n1, ptr(req2) if n2 else None, n2,
             ptr(alt["row_offset"]) if alt is not None else None, ptr(alt["count"]) if alt is not None else None,
             self.eps, self.seed, int(info["t"]), ptr(net), ptr(best), ptr(sel_row), ptr(sel_act), ro.stream)
        self._raw = (req1, req2, obs, sel_row, sel_act)
        return net, best

    @property
    def raw(self):
        """(departing agents, agents before a switch, observation rows, sel_row, sel_act) of the last step as the
        kernels left them, or None when nobody consulted the network."""
        return self._raw

    @property
    def last(self):
        """[(agents, observations, actions)] of the agents that consulted the network in the last step and had
        something to choose from (src/main.py:163-165, 194, 213), or None."""
        if self._last is None and self._raw is not None:
            req1, req2, obs, sel_row, sel_act = self._raw
            req = torch.cat([req1, req2]) if req2.numel() else req1
            ok = sel_row >= 0
            self._last = [(req[ok], obs[sel_row[ok].long()], sel_act[ok].long())]
        return self._last


class BatchedTrainer:
    def __init__(self, rollout: RolloutBatch, agent, replay_capacity=None, max_steps=200, start_eps=1.0, end_eps=0.005,
                 eps_decay=0.998, crash_penalty=False, switch2switch=False, max_learns_per_step=None, seed=0,
                 strict=False):
        self.ro = rollout
        self.agent = agent
        b = rollout.b
        if agent.height != 3 * b.Rp or agent.width != b.T1:
            raise ValueError("the network expects %dx%d inputs, the batch produces %dx%d"
                             % (agent.height, agent.width, 3 * b.Rp, b.T1))
        cap = replay_capacity or agent.args.buffer_size
        self.replay = DeviceReplay(cap, (3, b.plane_stride), rollout.device, seed=seed + 1)
        self.policy = DQNPolicy(agent, eps=start_eps, seed=seed)
        self.max_steps = max_steps
        self.eps, self.end_eps, self.eps_decay = start_eps, end_eps, eps_decay
        self.crash_penalty = crash_penalty
        self.switch2switch = switch2switch
        # curr_obs of src/main.py:165,213 (the observation an agent last acted on) and, for the
        # switch-to-switch mode, buffer_obs / buffer_rew of src/main.py:123,131: one slot per agent
        n = rollout.E * rollout.A
        dev = rollout.device
        self.curr_obs = self.buf_obs = None
        if crash_penalty or switch2switch:
            self.curr_obs = torch.zeros((n, 3, b.plane_stride), dtype=torch.uint8, device=dev)
            self.has_curr = torch.zeros((n,), dtype=torch.uint8, device=dev)
        if switch2switch:
            self.buf_obs = torch.zeros((n, 3, b.plane_stride), dtype=torch.uint8, device=dev)
            self.has_buf = torch.zeros((n,), dtype=torch.uint8, device=dev)
            self.buf_rew = torch.zeros((n,), dtype=torch.float64, device=dev)
        self.max_learns_per_step = max_learns_per_step
        self.strict = strict
        self.t_step = 0
        self.n_learn = 0
        self.n_transitions = 0
        self._rails_seen = 0

    def _learn(self):
        b = self.ro.b
        # every observation in the memory comes from this batch of environments: rails beyond its largest count are
        # empty, neither gathered nor computed
        # (of anything ever stored; with several ranks the largest over the ranks, agreed on in step(): the ranks' learn
        # steps must leave the same rows to the closed forms, or their gradients do not add up)
        rails = self._rails_seen = max(self._rails_seen, self.ro.rails_used())
        used = query("flm_dqn_learn_rails", self.agent.args.batch_size, b.Rp, b.T1, rails) * b.T1
        self.agent.learn_obs(self.replay.sample(self.agent.args.batch_size, used_bytes=used), self.agent.args.gamma,
                             b.plane_stride, rails_used=rails)
        self.n_learn += 1

    def step(self, train=True):
        ro = self.ro
        self.policy.eps = self.eps
        ro.step(self.policy)
        n_new = 0
        if train and self.curr_obs is None:
            # default feeding without per-agent observation slots (src/main.py:275-278): one pair of launches
            # stores the whole batch's transitions, next-state observations written straight into the ring
            raw = self.policy.raw
            if raw is not None:
                req1, req2, obs, sel_row, _ = raw
                n_new = self.replay.store_step(ro, req1, req2, sel_row, obs)
        elif train:
            # crash penalty (src/main.py:271-273) and switch-to-switch (src/main.py:280-296): the observation each
            # agent last acted on, and the one kept since its last "go", live in per-agent slots on the device
            raw = self.policy.raw
            req1, req2, obs, sel_row, _ = raw if raw is not None else (None, None, None, None, None)
            n_new = self.replay.store_step_slots(ro, req1, req2, sel_row, obs, self)
        due = 0
        if train:
            self.n_transitions += n_new
            # src/agent.py:57-67: one learn step per update_every stored transitions, once the memory is large enough
            due = (self.t_step + n_new) // self.agent.args.update_every
            self.t_step = (self.t_step + n_new) % self.agent.args.update_every
            if self.max_learns_per_step is not None:
                due = min(due, self.max_learns_per_step)
            if len(self.replay) <= self.agent.args.batch_size:
                due = 0
            # data parallel: every learn step all-reduces the gradient, so all ranks take the same number
            due = parallel.min_over_ranks(due, ro.device)
            if parallel.distributed():
                self._rails_seen = int(parallel.max_over_ranks(max(self._rails_seen, ro.rails_used()), ro.device))
            for _ in range(due):
                self._learn()
        return n_new

    def run_episode(self, train=True):
        """env.reset() + up to max_steps-1 steps (src/main.py:135-311) -> summary of the episode."""
        ro = self.ro
        ro.reset()
        if self.curr_obs is not None:
            self.has_curr.zero_()                          # curr_obs starts empty every episode (src/main.py:133)
        if self.buf_obs is not None:
            self.buf_rew.zero_()                           # src/main.py:131; buffer_obs itself outlives the episode (:123)
        total_reward = torch.zeros((ro.E,), dtype=torch.float64, device=ro.device)
        steps = 0
        for _ in range(self.max_steps - 1):
            self.step(train=train)
            total_reward += ro.reward.sum(dim=1)
            steps += 1
            if parallel.min_over_ranks(int(ro.all_done.all().item()), ro.device):   # every rank's environments are done
                break
        self.eps = max(self.end_eps, self.eps_decay * self.eps)
        done_frac = ro.done.float().mean().item()
        flagged, overflowed = ro.flagged_agents()
        if overflowed:
            # not a case of the reference: this batch's own capacities were too small, the agents concerned stood still
            raise OverflowError("%d agents had more than cap_alt=%d alternatives or a path longer than %d elements"
                                % (overflowed, ro.cap_alt, ro.Lp))
        if flagged and self.strict:
            raise RuntimeError("%d agents' paths or alternatives raised (IndexError / KeyError / AssertionError in "
                               "the reference's terms)" % flagged)
        return dict(steps=steps, env_steps=steps * ro.E, done_agents=done_frac,
                    mean_reward=float(total_reward.mean().item()), eps=self.eps, learns=self.n_learn,
                    transitions=self.n_transitions, agents_flagged=flagged)
