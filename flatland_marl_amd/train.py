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

Differences from the sequential script, all inherent to stepping many environments together: the learn
steps of one rollout step sample a memory that already holds all of that step's transitions; random numbers
come from one device generator; the switch-to-switch replay mode (src/main.py:280-296) is not implemented.
"""
import numpy as np
import torch

from . import parallel
from ._lib import call, ptr
from .rollout import RolloutBatch


def obs_to_f32(obs_u8, Rp, T1, plane_stride, out=None):
    """uint8 observations [n,3,plane_stride] -> the network's input [n,3*Rp,T1] float32."""
    n = obs_u8.shape[0]
    if out is None:
        out = torch.empty((n, 3 * Rp, T1), dtype=torch.float32, device=obs_u8.device)
    call("flm_obs_to_f32", ptr(obs_u8), n, Rp, T1, plane_stride, ptr(out),
         torch.cuda.current_stream(obs_u8.device).cuda_stream)
    return out


class DQNPolicy:
    """policy(rollout, info) for RolloutBatch.step; keeps what the replay memory needs in `last`: a list of
    (agents e*A+a, their observations as shown to the network, the actions chosen) per group of agents."""

    def __init__(self, agent, eps=0.0, seed=0, chunk_bytes=2 << 30):
        self.agent = agent
        self.eps = float(eps)
        self.chunk_bytes = int(chunk_bytes)
        self.gen = None
        self.seed = seed
        self.last = None

    def _rand(self, n, dev):
        if self.gen is None:
            self.gen = torch.Generator(device=dev)
            self.gen.manual_seed(self.seed)
        return torch.rand(n, generator=self.gen, device=dev)

    def q_values(self, ro, obs_u8):
        b = ro.b
        n = obs_u8.shape[0]
        q = torch.empty((n, 2), dtype=torch.float32, device=obs_u8.device)
        per = 64 * 3 * b.Rp * 4                            # the convolution output dominates the workspace
        chunk = max(1, self.chunk_bytes // per)
        net = self.agent.qnetwork_local
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            q[lo:hi] = net.forward_obs(obs_u8[lo:hi], b.plane_stride)
        return q

    def __call__(self, ro, info):
        E, A, dev = ro.E, ro.A, ro.device
        net = torch.zeros((E, A), dtype=torch.int32, device=dev)
        best = torch.zeros((E, A), dtype=torch.int32, device=dev)
        reqs, obss, acts = [], [], []
        req1 = info["req_ready"]
        if req1.numel():                                   # src/main.py:164-172
            obs1 = ro.consult_observations(req1)
            q1 = self.q_values(ro, obs1)
            n1 = req1.numel()
            greedy = q1.argmax(dim=1)
            explore = self._rand(n1, dev) <= self.eps
            coin = (self._rand(n1, dev) < 0.5).long()
            a1 = torch.where(explore, coin, greedy)
            net.view(-1)[req1.long()] = a1.to(torch.int32)
            reqs.append(req1)
            obss.append(obs1)
            acts.append(a1)
        alt = info["alt"]
        if alt is not None:                                # src/main.py:193-213
            req2 = info["req_switch"]
            cnt = alt["count"].long()
            K = ro.cap_alt
            used = torch.arange(K, device=dev)[None, :] < cnt[:, None]
            ii, kk = torch.nonzero(used, as_tuple=True)
            if ii.numel():
                obs2 = ro.consult_observations(req2[ii].contiguous(), alt["cols"][ii, kk].contiguous())
                q2 = self.q_values(ro, obs2)
                qfull = torch.full((req2.numel(), K, 2), float("-inf"), device=dev)
                qfull[ii, kk] = q2
                arg = qfull.view(req2.numel(), -1).argmax(dim=1)
                g_best, g_net = arg // 2, arg % 2
                n2 = req2.numel()
                explore = self._rand(n2, dev) <= self.eps
                r_net = (self._rand(n2, dev) < 0.5).long()
                r_best = torch.minimum((self._rand(n2, dev) * cnt.clamp(min=1)).long(), cnt.clamp(min=1) - 1)
                a2 = torch.where(explore, r_net, g_net)
                b2 = torch.where(explore, r_best, g_best)
                has = cnt > 0
                net.view(-1)[req2.long()] = torch.where(has, a2, torch.zeros_like(a2)).to(torch.int32)
                best.view(-1)[req2.long()] = torch.where(has, b2, torch.zeros_like(b2)).to(torch.int32)
                row = torch.full((n2, K), -1, dtype=torch.long, device=dev)
                row[ii, kk] = torch.arange(ii.numel(), device=dev)
                sel = row[torch.arange(n2, device=dev), b2.clamp(max=K - 1)]
                keep = has & (sel >= 0)
                reqs.append(req2[keep])
                obss.append(obs2[sel[keep]])
                acts.append(a2[keep])
        # one entry per group of consulting agents (departing, in front of a switch): (agents, observations, actions)
        self.last = list(zip(reqs, obss, acts)) if reqs else None
        return net, best


class DeviceReplay:
    def __init__(self, capacity, plane_stride, device, seed=0):
        self.capacity = int(capacity)
        self.ps = plane_stride
        self.state = torch.empty((self.capacity, 3, plane_stride), dtype=torch.uint8, device=device)
        self.next_state = torch.empty((self.capacity, 3, plane_stride), dtype=torch.uint8, device=device)
        self.action = torch.zeros((self.capacity,), dtype=torch.int64, device=device)
        self.reward = torch.zeros((self.capacity,), dtype=torch.float32, device=device)
        self.done = torch.zeros((self.capacity,), dtype=torch.float32, device=device)
        self.size = 0
        self.head = 0
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)

    def __len__(self):
        return self.size

    def add(self, state, action, reward, next_state, done):
        """Append n transitions (oldest entries are overwritten, as with deque(maxlen))."""
        n = state.shape[0]
        if n > self.capacity:
            state, action, reward, next_state, done = (x[-self.capacity:] for x in (state, action, reward, next_state, done))
            n = self.capacity
        # the ring is filled with at most two contiguous copies
        first = min(n, self.capacity - self.head)
        for dst, src in ((self.state, state), (self.next_state, next_state), (self.action, action.to(torch.int64)),
                         (self.reward, reward.to(torch.float32)), (self.done, done.to(torch.float32))):
            dst[self.head:self.head + first].copy_(src[:first])
            if n > first:
                dst[:n - first].copy_(src[first:])
        self.head = (self.head + n) % self.capacity
        self.size = min(self.capacity, self.size + n)

    def sample(self, batch_size):
        idx = torch.randperm(self.size, generator=self.gen, device=self.state.device)[:batch_size]
        return self.state[idx], self.action[idx], self.reward[idx], self.next_state[idx], self.done[idx]


class BatchedTrainer:
    def __init__(self, rollout: RolloutBatch, agent, replay_capacity=None, max_steps=200, start_eps=1.0, end_eps=0.005,
                 eps_decay=0.998, crash_penalty=False, max_learns_per_step=None, seed=0):
        self.ro = rollout
        self.agent = agent
        b = rollout.b
        if agent.height != 3 * b.Rp or agent.width != b.T1:
            raise ValueError("the network expects %dx%d inputs, the batch produces %dx%d"
                             % (agent.height, agent.width, 3 * b.Rp, b.T1))
        cap = replay_capacity or agent.args.buffer_size
        self.replay = DeviceReplay(cap, b.plane_stride, rollout.device, seed=seed + 1)
        self.policy = DQNPolicy(agent, eps=start_eps, seed=seed)
        self.max_steps = max_steps
        self.eps, self.end_eps, self.eps_decay = start_eps, end_eps, eps_decay
        self.crash_penalty = crash_penalty
        self.max_learns_per_step = max_learns_per_step
        self.t_step = 0
        self.n_learn = 0
        self.n_transitions = 0

    def _learn(self):
        b = self.ro.b
        s, a, r, sn, d = self.replay.sample(self.agent.args.batch_size)
        B = s.shape[0]
        sf = obs_to_f32(s.contiguous(), b.Rp, b.T1, b.plane_stride).view(B, 1, 3 * b.Rp, b.T1)
        snf = obs_to_f32(sn.contiguous(), b.Rp, b.T1, b.plane_stride).view(B, 1, 3 * b.Rp, b.T1)
        self.agent.learn((sf, a, r, snf, d), self.agent.args.gamma)
        self.n_learn += 1

    def step(self, train=True):
        ro = self.ro
        self.policy.eps = self.eps
        ro.step(self.policy)
        last = self.policy.last
        n_new = 0
        if train and last is not None:
            for req32, obs, _ in last:
                req = req32.long()
                if req.numel() == 0:
                    continue
                # the observation after the step: the agent's own plane against everybody's (src/main.py:277)
                next_obs = ro.consult_observations(req32.contiguous())
                reward = ro.reward.view(-1)[req]
                done = ro.done.view(-1)[req]
                net_out = ro.net.view(-1)[req]             # the decision after the crash checks (src/main.py:180, 225)
                self.replay.add(obs, net_out, reward, next_obs, done)
                n_new += req.numel()
                if self.crash_penalty:                     # src/main.py:271-273, for the agents that consulted this step
                    cr = ro.crash.view(-1)[req] != 0
                    k = int(cr.sum().item())
                    if k:
                        o = obs[cr]
                        self.replay.add(o, torch.ones(k, dtype=torch.int64, device=ro.device),
                                        torch.full((k,), -100.0, device=ro.device), o, torch.ones(k, device=ro.device))
                        n_new += k
            self.n_transitions += n_new
            # src/agent.py:57-67: one learn step per update_every stored transitions, once the memory is large enough
            due = (self.t_step + n_new) // self.agent.args.update_every
            self.t_step = (self.t_step + n_new) % self.agent.args.update_every
            if self.max_learns_per_step is not None:
                due = min(due, self.max_learns_per_step)
            if len(self.replay) <= self.agent.args.batch_size:
                due = 0
        else:
            due = 0
        if train:
            # data parallel: every learn step all-reduces the gradient, so all ranks take the same number
            due = parallel.min_over_ranks(due, ro.device)
            for _ in range(due):
                self._learn()
        return n_new

    def run_episode(self, train=True):
        """env.reset() + up to max_steps-1 steps (src/main.py:135-311) -> summary of the episode."""
        ro = self.ro
        ro.reset()
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
        return dict(steps=steps, env_steps=steps * ro.E, done_agents=done_frac,
                    mean_reward=float(total_reward.mean().item()), eps=self.eps, learns=self.n_learn,
                    transitions=self.n_transitions)
