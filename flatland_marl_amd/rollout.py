"""Batched rollout: the per-step loop of the reference's training script (src/main.py:142-252) for E
environments at once, state resident on the GPU.

The occupancy planes live in column form ([E,A,T1] int16, see ObservationBatch.columns): every rule of the
loop (shift, delay, crash checks) moves or reads whole columns.  One step is

    classify   -> which of the five cases of src/main.py:146-249 each agent is in
    altpaths   -> alternatives (paths + planes) of the agents in front of a switch
    policy     -> stop/go and the chosen alternative (the network of src/main.py:168-172, 197-208, or any
                  other callable; it may ask for the observations of the consulting agents)
    apply      -> the sequential per-agent bookkeeping, one wave per environment
    env step   -> the environment advances (stand-in for flatland's RailEnv.step, with its malfunction
                  process when asked for)

Dense observations are only produced for the agents that consult the policy.
"""
import numpy as np
import torch

from ._lib import call, ptr
from .batch import ObservationBatch

MAX_REQ = 65535      # requests per launch of the request kernels


class _AltViews(dict):
    """What RolloutBatch.alternatives returns: views of the pooled tables for n requests, made on first use (a step
    touches a few of them, and slicing all of them costs more host time than the kernels leave)."""
    _PER_REQUEST = ("count", "len", "cell", "meta", "err", "cols", "col_err")

    def __init__(self, pool, n, n_rows, req):
        super().__init__(req=req)
        self._pool, self._n, self._n_rows = pool, n, n_rows

    def __missing__(self, key):
        if key in self._PER_REQUEST:
            v = self._pool[key][:self._n]
        elif key == "row_offset":
            v = self._pool[key][:self._n + 1]
        elif key in ("row_req", "row_own"):
            v = self._pool[key][:self._n_rows]
        else:
            raise KeyError(key)
        self[key] = v
        return v


class RolloutBatch:
    def __init__(self, batch: ObservationBatch, speed, initial_direction=None, cap_alt=8, max_alt_depth=500,
                 malfunction=None):
        """malfunction = dict(rate, min_duration, max_duration, seed[, env_seeds]): the parameters of the
        malfunction_from_params generator of src/main.py:72-77 (rate 0 = never; the script's defaults are
        2000, 0, 0).  Environment e draws from the seed env_seeds[e] (default: a hash of seed and e)."""
        self.b = b = batch
        E, A, dev = b.E, b.A, b.device
        self.E, self.A, self.device = E, A, dev
        self.cap_alt = int(cap_alt)
        # The reference asks for alternative paths of up to 500 moves (src/predictions.py:75).  An agent
        # consumes at most one path element per environment step and a plane shows depth+1 steps, so in an
        # episode of at most S steps a smaller max_alt_depth >= max(S, depth + 1) gives the same rollout.
        self.max_alt_depth = int(max_alt_depth)
        self.Lp = self.max_alt_depth + 40
        if self.Lp < b.T1 + 1:
            # reset() copies the shortest paths (up to depth + 2 elements) into rows of Lp elements
            raise ValueError("max_alt_depth + 40 = %d path elements cannot hold a shortest path of depth + 2 = %d"
                             % (self.Lp, b.T1 + 1))
        self.speed = torch.as_tensor(np.asarray(speed, np.float64).reshape(E, A), device=dev)
        self.init_dir = b.direction.clone() if initial_direction is None else \
            torch.as_tensor(np.asarray(initial_direction, np.int32).reshape(E, A), device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.status = torch.zeros((E, A), **i32)
        self.pos = torch.full((E, A), -1, **i32)             # cell on the grid, -1 = not on the grid
        self.heading = self.init_dir.clone()
        self.moving = torch.zeros((E, A), **i32)
        self.fraction = torch.zeros((E, A), dtype=torch.float64, device=dev)
        self.exit_action = torch.zeros((E, A), **i32)
        self.done = torch.zeros((E, A), **i32)
        self.all_done = torch.zeros((E,), **i32)
        self.reward = torch.zeros((E, A), dtype=torch.float64, device=dev)
        self.action_required = torch.ones((E, A), **i32)
        self.path_cell = torch.zeros((E, A, self.Lp), **i32)
        self.path_meta = torch.zeros((E, A, self.Lp), **i32)
        self.path_head = torch.zeros((E, A), **i32)
        self.path_len = torch.zeros((E, A), **i32)
        self.branch = torch.zeros((E, A), **i32)
        self.before_switch = torch.zeros((E, A), **i32)
        self.action = torch.zeros((E, A), **i32)
        self.crash = torch.zeros((E, A), **i32)
        self.net = torch.zeros((E, A), **i32)
        self.alt_slot = torch.full((E, A), -1, **i32)
        self.agent_err = torch.zeros((E, A), **i32)          # sticky: first error code seen for the agent
        self.ready_list = torch.zeros((E * A,), **i32)      # agents in case 1 / case 2 this step, ascending
        self.switch_list = torch.zeros((E * A,), **i32)
        # departing agents, agents before a switch, network rows: the kernels write these three numbers straight into
        # pinned host memory, the host reads them after an event (no copy in between)
        self.req_counts = torch.zeros((3,), dtype=torch.int32).pin_memory()
        self._counts_event = torch.cuda.Event()
        self._req_ws = torch.zeros((2 * ((E * A + 1023) // 1024),), **i32)
        self._dummy_i32 = torch.zeros((1,), **i32)
        self._dummy_i16 = torch.zeros((1,), dtype=torch.int16, device=dev)
        self.cols = None
        self._alt_pool = None
        self._rails_used = None             # (of the batch's graphs: asked for again after a reset)
        self._alt_guess = 0                 # agents before a switch expected in the coming step (0: not yet known)
        self.t = 0
        self.malfunction = self.malfunction_moving = self.env_seed = None
        self.malf_threshold, self.malf_min, self.malf_max = 0, 0, 0
        if malfunction and malfunction.get("rate", 0) > 0:
            self.malf_threshold = min(0xFFFFFFFF, int((1.0 - np.exp(-1.0 / malfunction["rate"])) * 4294967296.0))
            self.malf_min, self.malf_max = int(malfunction["min_duration"]), int(malfunction["max_duration"])
            seeds = malfunction.get("env_seeds")
            if seeds is None:
                x = (np.arange(E, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(int(malfunction.get("seed", 0)))
                     ) & np.uint64(0xFFFFFFFF)
                x ^= x >> np.uint64(16)
                x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
                x ^= x >> np.uint64(13)
                seeds = x
            seeds = np.asarray(seeds, np.uint64).reshape(E).astype(np.int64)
            # uint32 bit patterns in an int32 tensor
            self.env_seed = torch.as_tensor(np.where(seeds >= 2 ** 31, seeds - 2 ** 32, seeds).astype(np.int32), device=dev)
            self.malfunction = torch.zeros((E, A), **i32)
            self.malfunction_moving = torch.zeros((E, A), **i32)

    @property
    def stream(self):
        return self.b.stream.cuda_stream

    # ------------------------------------------------------------------ reset
    def reset(self):
        """env.reset(): every agent ready to depart, shortest paths and planes of all agents (get_many)."""
        b = self.b
        self.status.zero_()
        self.pos.fill_(-1)
        self.heading.copy_(self.init_dir)
        self.moving.zero_()
        self.fraction.zero_()
        self.exit_action.zero_()
        self.done.zero_()
        self.all_done.zero_()
        self.action_required.fill_(1)
        self.agent_err.zero_()
        if self.malfunction is not None:
            self.malfunction.zero_()
            self.malfunction_moving.zero_()
        b.status.zero_()
        b.pos_cell.copy_(b.start_cell)
        b.direction.copy_(self.init_dir)
        if b.R is None:
            b.build_graph(sync=True)
        if getattr(b, "dist", None) is None:
            b.compute_distances()
        b.shortest_paths()
        self._rails_used = None
        self._alt_guess = 0
        self.cols = b.columns(shift=1)
        n = min(b.Lcap, self.Lp)
        self.path_cell[:, :, :n] = b.path_cell[:, :, :n]
        self.path_meta[:, :, :n] = b.path_meta[:, :, :n]
        self.path_len.copy_(b.path_len)
        self.path_head.zero_()
        self.agent_err.copy_(b.agent_err)
        self.t = 0
        return self

    # ------------------------------------------------------------------ pieces of a step
    def classify(self):
        b = self.b
        call("flm_rollout_classify", ptr(self.status), ptr(self.pos), ptr(self.action_required), ptr(self.path_cell),
             ptr(self.path_meta), ptr(self.path_head), ptr(self.path_len), ptr(b.cell_rail), self.E, self.A, b.H, b.W,
             self.Lp, ptr(self.branch), ptr(self.before_switch), self.stream)
        return self.branch

    def _alt_tables(self):
        """Pointers of the tables flm_altpaths reads: gathered before the host waits for the step's sizes, so that the
        launch follows the wait at once."""
        b = self.b
        return (ptr(b.grid), ptr(b.cell_idx), ptr(b.dist), ptr(b.succ_table()), ptr(b.choice_table()), ptr(b.idx_cell),
                ptr(b.tslot), ptr(b.node_id)), (ptr(b.pos_cell), ptr(b.direction), ptr(b.target_cell))

    def _alt_pool_for(self, n, n_ready):
        # the tables live in buffers that only ever grow: a step allocates nothing once they have reached the
        # largest number of agents in front of a switch seen so far
        b, K, pool = self.b, self.cap_alt, self._alt_pool
        if pool is None or pool["cap"] < n or pool["cap_rows"] < n_ready + n * K:
            i32 = dict(dtype=torch.int32, device=self.device)
            cap = max(n, 64, 0 if pool is None else pool["cap"] * 3 // 2)
            cap_rows = max(n_ready + n * K, self.E * self.A + cap * K)
            pool = self._alt_pool = dict(
                cap=cap, cap_rows=cap_rows, count=torch.empty((cap,), **i32), len=torch.zeros((cap, K), **i32),
                cell=torch.empty((cap, K, self.Lp), **i32), meta=torch.empty((cap, K, self.Lp), **i32),
                err=torch.empty((cap,), **i32), cols=torch.empty((cap, K, b.T1), dtype=torch.int16, device=self.device),
                col_err=torch.empty((cap, K), **i32), row_offset=torch.empty((cap + 1,), **i32),
                row_req=torch.empty((cap_rows,), **i32), row_own=torch.empty((cap_rows,), **i32))
            pool["agent_rep"] = torch.empty((cap * K,), **i32)
            pool["base"] = tuple(pool[k].data_ptr() for k in ("count", "len", "cell", "meta", "err"))
            pool["ptrs"] = {k: pool[k].data_ptr() for k in ("cols", "col_err", "row_offset", "row_req", "row_own")}
        return pool

    def alternatives(self, req, ready=None, tables=None, launched=0):
        """Alternative paths and their planes (column form) for the agents req[n] (flat e*A+a, all ACTIVE):
        get_altmaps, src/rail_observations.py:97-111.  The batch's pos_cell / direction / status must hold
        the agents' current state (step() sees to it).  ready: the departing agents of the same step; the
        rows the network has to evaluate (ready first, then every alternative) are listed in row_req /
        row_own, their number in req_counts[2].  launched: flm_altpaths_counted has already been enqueued for the
        first so many requests (step() does that before it knows their number), into the current pool.
        Returns a dict of views of pooled buffers (count, len, cell, meta, err, cols, col_err, row_offset, row_req,
        row_own) and req; the views are made when they are first asked for."""
        b = self.b
        n, K = req.numel(), self.cap_alt
        n_ready = 0 if ready is None else ready.numel()
        before = self._alt_pool
        pool = self._alt_pool_for(n, n_ready)
        if pool is not before:
            launched = 0                                    # the pool had to grow: what was enqueued wrote into the old one
        fixed, state = tables if tables is not None else self._alt_tables()
        p_count, p_len, p_cell, p_meta, p_err = pool["base"]
        p_req, st, Lp = ptr(req), self.stream, self.Lp
        p_rep = pool["agent_rep"].data_ptr()
        if tables is not None:
            # inside a step: the kernel that counted the requests is ahead on the stream, the launch takes their number
            # from there and also writes the per-path agent list
            for lo in range(min(launched, n), n, MAX_REQ):
                call("flm_altpaths_counted", *fixed, p_req + 4 * lo, min(n, lo + MAX_REQ) - lo, self.req_counts.data_ptr() + 4, lo,
                     p_rep + 4 * K * lo, *state, self.E, self.A, b.H, b.W, b.cap_cells, self.max_alt_depth, K, Lp,
                     p_count + 4 * lo, p_len + 4 * K * lo, p_cell + 4 * K * Lp * lo, p_meta + 4 * K * Lp * lo, p_err + 4 * lo, st)
        else:
            for lo in range(0, n, MAX_REQ):
                call("flm_altpaths", *fixed, p_req + 4 * lo, min(n, lo + MAX_REQ) - lo, *state, self.E, self.A, b.H, b.W,
                     b.cap_cells, self.max_alt_depth, K, Lp, p_count + 4 * lo, p_len + 4 * K * lo, p_cell + 4 * K * Lp * lo,
                     p_meta + 4 * K * Lp * lo, p_err + 4 * lo, st)
            # one path per (request, alternative); unused alternatives have length 0
            pool["agent_rep"][:n * K].view(n, K).copy_(req[:, None].expand(n, K))
        ptrs = pool["ptrs"]
        call("flm_columns", p_cell, p_len, p_rep, ptr(b.pos_cell), ptr(b.target_cell),
             ptr(b.tpc), ptr(b.node_id), ptr(b.cell_rail), ptr(b.edge_info), self.E, self.A, n * K, b.H, b.W, b.capR,
             b.Rp, b.T1, self.Lp, 0, ptrs["cols"], ptrs["col_err"], self.stream)
        # an alternative whose path cannot be mapped on the rail graph raises in the reference; here the agent
        # is flagged and treated as having no alternatives
        call("flm_rollout_alt_finish", p_count, p_err, ptrs["col_err"], p_req, n, K,
             ptr(ready) if n_ready else None, n_ready, ptr(self.agent_err), ptrs["row_offset"],
             ptrs["row_req"], ptrs["row_own"], ptr(self.req_counts), self.stream)
        return _AltViews(pool, n, n_ready + n * K, req)

    def consult_observations(self, req, own_cols=None, out=None, own_row=None):
        """[n,3,plane_stride] uint8 observations of the agents req[n] against the current planes
        (get_obs, src/preprocessing.py:61-74); own_cols replaces the planes shown as their own: row g of
        own_cols [n,T1] for request g, or row own_row[g] of own_cols [.,T1] (negative: the agent's own)."""
        b = self.b
        n = req.numel()
        if out is None:
            out = torch.empty((n, 3, b.plane_stride), dtype=torch.uint8, device=self.device)
        p_req, p_out, p_cols = ptr(req), ptr(out), ptr(own_cols) if own_cols is not None else None
        p_row = ptr(own_row) if own_row is not None else None
        for lo in range(0, n, MAX_REQ):
            if p_row is not None:
                oc, orow = p_cols, p_row + 4 * lo
            else:
                oc, orow = (p_cols + 2 * b.T1 * lo if p_cols is not None else None), None
            call("flm_observations_req_columns", ptr(self.cols), p_req + 4 * lo, oc, orow, min(n, lo + MAX_REQ) - lo, self.E,
                 self.A, b.Rp, b.T1, b.plane_stride, p_out + 3 * b.plane_stride * lo, self.stream)
        return out

    def consult_observations_counted(self, req, own_cols, own_row, out, n_max, plane_limit=0):
        """consult_observations for the first req_counts[2] of the n_max requests listed in req, enqueued before the
        host knows that number (the kernel that counted them is ahead on the stream); out holds n_max rows.
        plane_limit > 0: only so many bytes of every plane are written (plane_bytes_used()); out must hold zeros
        behind them."""
        b = self.b
        if n_max > MAX_REQ:
            raise ValueError("at most %d requests per launch" % MAX_REQ)
        call("flm_observations_req_columns_counted", ptr(self.cols), ptr(req), ptr(own_cols), ptr(own_row), n_max,
             self.req_counts.data_ptr() + 8, int(plane_limit), self.E, self.A, b.Rp, b.T1, b.plane_stride, ptr(out), self.stream)

    def plane_bytes_used(self):
        """Bytes at the start of every observation plane that can be other than zero: the rails the batch's
        environments have, rounded up to 16 bytes."""
        return min(self.b.plane_stride, (self.rails_used() * self.b.T1 + 15) // 16 * 16)

    def apply(self, net_in, best, alt):
        b = self.b
        a = alt if alt is not None else dict(count=self._dummy_i32, len=self._dummy_i32, cell=self._dummy_i32,
                                             meta=self._dummy_i32, cols=self._dummy_i16)
        call("flm_rollout_apply", ptr(self.cols), ptr(self.status), ptr(b.tpc), ptr(self.pos), ptr(b.start_cell),
             ptr(self.branch), ptr(net_in), ptr(self.alt_slot), ptr(best), ptr(a["count"]), ptr(a["len"]),
             ptr(a["cell"]), ptr(a["meta"]), ptr(a["cols"]), ptr(self.path_cell), ptr(self.path_meta),
             ptr(self.path_head), ptr(self.path_len), self.E, self.A, b.W, b.T1, self.Lp, self.cap_alt,
             ptr(self.action), ptr(self.crash), ptr(self.net), self.stream)

    def env_step(self):
        b = self.b
        call("flm_env_step", ptr(b.grid), ptr(self.action), ptr(b.start_cell), ptr(self.init_dir), ptr(b.target_cell),
             ptr(self.speed), ptr(self.status), ptr(self.pos), ptr(self.heading), ptr(self.moving), ptr(self.fraction),
             ptr(self.exit_action), ptr(self.done), self.E, self.A, b.H, b.W, ptr(self.reward),
             ptr(self.action_required), ptr(self.all_done),
             ptr(self.malfunction) if self.malfunction is not None else None,
             ptr(self.malfunction_moving) if self.malfunction is not None else None,
             ptr(self.env_seed) if self.malfunction is not None else None,
             self.malf_threshold, self.malf_min, self.malf_max, self.t, self.stream)

    # ------------------------------------------------------------------ one step
    def step(self, policy):
        """One pass of src/main.py:142-252 over every environment.

        policy(rollout, info) -> (net [E,A] int32: 1 = go, best [E,A] int32: chosen alternative).  info holds
        branch [E,A], req_ready (flat indices of the agents about to depart), req_switch (agents before a
        switch), alt (their alternatives: count, len, cell, meta, cols, and the rows for the network:
        row_offset, row_req, row_own), net / best (zeroed [E,A] buffers a policy may fill and return) and t
        (the step number).  The policy
        sees the planes as they were when the step began (src/main.py:144)."""
        b = self.b
        self.classify()
        # the consulting agents (ordered lists), cleared decisions, and the batch's view of the agents
        if getattr(self, "_net", None) is None:
            self._net = torch.empty((self.E, self.A), dtype=torch.int32, device=self.device)
            self._best = torch.empty((self.E, self.A), dtype=torch.int32, device=self.device)
        call("flm_rollout_requests", ptr(self.branch), ptr(self.status), ptr(self.pos), ptr(self.heading),
             ptr(b.start_cell), self.E, self.A, ptr(self.ready_list), ptr(self.switch_list), ptr(self.req_counts),
             ptr(self.alt_slot), ptr(self._net), ptr(self._best), ptr(b.pos_cell), ptr(b.direction), ptr(b.status),
             ptr(self._req_ws), self.stream)
        # what needs no sizes is done before the host waits for them: the device is still busy with the kernels above
        tables = self._alt_tables()
        prepare = getattr(policy, "prepare", None)
        if prepare is not None:
            prepare(self)
        # ... and the alternative paths are enqueued for about as many agents as asked in the last step: the kernel takes
        # the real number from where the kernel above left it, the host learns it while the paths are being followed,
        # and whoever was not covered gets a launch of their own
        self.mark_counts()
        launched = min(self._alt_guess, self.E * self.A, MAX_REQ)
        if launched:
            pool = self._alt_pool_for(launched, 0)
            p_count, p_len, p_cell, p_meta, p_err = pool["base"]
            call("flm_altpaths_counted", *tables[0], ptr(self.switch_list), launched, self.req_counts.data_ptr() + 4, 0,
                 pool["agent_rep"].data_ptr(), *tables[1], self.E, self.A, b.H, b.W, b.cap_cells, self.max_alt_depth,
                 self.cap_alt, self.Lp, p_count, p_len, p_cell, p_meta, p_err, self.stream)
        n_ready, n_switch = self.wait_counts()[:2]             # the host waits for the step's sizes
        self._alt_guess = n_switch + n_switch // 4 + 32 if n_switch else 0
        req_ready, req_switch = self.ready_list[:n_ready], self.switch_list[:n_switch]
        alt = self.alternatives(req_switch, req_ready, tables, launched) if n_switch else None
        info = dict(t=self.t, branch=self.branch, req_ready=req_ready, req_switch=req_switch, alt=alt,
                    net=self._net, best=self._best)
        net_in, best = policy(self, info)
        self.apply(net_in, best, alt)
        self.env_step()
        self.t += 1
        info["net_in"], info["best"] = net_in, best
        return info

    def rails_used(self):
        """The largest number of rails of any environment of the batch: rows from there on are empty in every plane
        of every observation (the padding up to max_rails)."""
        if self._rails_used is None:
            self._rails_used = int(np.asarray(self.b.R).max())
        return self._rails_used

    def mark_counts(self):
        """The point of the stream at which wait_counts() reads the step's sizes: behind the kernels enqueued so far."""
        self._counts_event.record(self.b.stream)

    def wait_counts(self):
        self._counts_event.synchronize()
        return self.req_counts.tolist()

    def read_counts(self):
        """[departing agents, agents before a switch, network rows] as the kernels enqueued so far leave them."""
        self.mark_counts()
        return self.wait_counts()

    def flagged_agents(self):
        """(agents whose path or alternatives raised in the reference's terms, agents that hit a capacity of this
        batch: more than cap_alt alternatives or a path longer than Lp) since the last reset.  A flagged agent is
        treated as having no alternatives and stops in front of its switch; the second kind has no counterpart
        in the reference and means cap_alt / max_alt_depth are too small for the maps."""
        err = self.agent_err
        return int((err != 0).sum().item()), int((err == 5).sum().item())

    # ------------------------------------------------------------------ host views (tests)
    def maps_numpy(self, e=0, R=None):
        """Dense [A,R,T1] int8 planes of one environment, rebuilt from the columns."""
        b = self.b
        R = int(b.R[e]) if R is None else R
        cols = self.cols[e].cpu().numpy().astype(np.int64)
        out = np.zeros((self.A, R, b.T1), np.int8)
        a, t = np.nonzero(cols >= 0)
        c = cols[a, t]
        out[a, c >> 1, t] = np.where(c & 1, 1, -1)
        return out


def hash_policy(rollout, info):
    """Deterministic stand-in for the network (tests): go 7 times out of 8, alternative = hash % n_alt;
    the same function of (step, agent) for every environment."""
    A, dev = rollout.A, rollout.device
    net = np.zeros(A, np.int32)
    sel = np.zeros(A, np.int64)
    for a in range(A):
        x = (info["t"] * 7919 + a * 104729 + 12345) & 0x7FFFFFFF
        x = (x * 1103515245 + 12345) & 0x7FFFFFFF
        x = (x * 1103515245 + 12345) & 0x7FFFFFFF
        net[a] = 1 if ((x >> 16) % 8) != 0 else 0
        sel[a] = (x >> 8) & 0xFFFF
    net_in = torch.as_tensor(net, device=dev)[None, :].expand(rollout.E, A).contiguous()
    best = torch.zeros((rollout.E, A), dtype=torch.int32, device=dev)
    alt = info["alt"]
    if alt is not None:
        req = info["req_switch"].long()
        n_alt = alt["count"].long()
        s = torch.as_tensor(sel, device=dev)[req % A]
        best.view(-1)[req] = torch.where(n_alt > 0, s % torch.clamp(n_alt, min=1), torch.zeros_like(s)).to(torch.int32)
    return net_in, best
