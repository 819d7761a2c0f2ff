"""Replay memory in device memory (the reference keeps a deque of float64 arrays on the host,
src/agent.py:134-198): a ring of transitions, uniform sampling without replacement by flm_replay_sample, the
sampled rows gathered into buffers that are allocated once."""
import torch

from ._lib import call, ptr, query


class DeviceReplay:
    """capacity transitions of (state row, action, reward, next-state row, done).  A row is row_shape elements of
    `dtype`: the uint8 observation [3, plane_stride] of the batched trainer, or the float32 [H3*T1] state of the
    drop-in agent; its size in bytes must be a multiple of 16."""

    def __init__(self, capacity, row_shape, device, seed=0, dtype=torch.uint8):
        self.capacity = int(capacity)
        self.row_shape = tuple(int(x) for x in (row_shape if hasattr(row_shape, "__len__") else (row_shape,)))
        self.dtype = dtype
        n = 1
        for x in self.row_shape:
            n *= x
        self.row_bytes = n * torch.empty((), dtype=dtype).element_size()
        if self.row_bytes % 16:
            raise ValueError("a replay row of %d bytes is not a multiple of 16" % self.row_bytes)
        self.device = torch.device(device)
        self.state = torch.empty((self.capacity,) + self.row_shape, dtype=dtype, device=self.device)
        self.next_state = torch.empty((self.capacity,) + self.row_shape, dtype=dtype, device=self.device)
        self.action = torch.zeros((self.capacity,), dtype=torch.int64, device=self.device)
        self.reward = torch.zeros((self.capacity,), dtype=torch.float32, device=self.device)
        self.done = torch.zeros((self.capacity,), dtype=torch.float32, device=self.device)
        self.size = 0
        self.head = 0
        self.seed = int(seed) & 0xFFFFFFFF
        self.draws = 0
        self._out = None

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

    def store_step(self, ro, req_ready, req_switch, sel_row, obs):
        """The default feeding mode of the training script (src/main.py:275-278) for one rollout step of every
        environment: the agents req_ready + req_switch that acted on row sel_row[i] >= 0 of obs [rows,3,plane_stride]
        are stored with the decision left by the crash checks, the step's reward and done flag, and their
        observation on the planes after the step, which flm_observations_req_columns_rows writes straight into
        the ring.  Returns the number of transitions (one host read-back)."""
        n1, n2 = req_ready.numel(), req_switch.numel()
        n = n1 + n2
        dev = self.device
        if getattr(self, "_slot", None) is None or self._slot.numel() < n:
            cap = max(n, 1024)
            self._slot = torch.empty((cap,), dtype=torch.int32, device=dev)
            self._req = torch.empty((cap,), dtype=torch.int32, device=dev)
            # the kernels write the step's number of transitions into pinned host memory; the host reads it after an event
            self._count = torch.zeros((1,), dtype=torch.int32).pin_memory()
            self._count_event = torch.cuda.Event()
        b = ro.b
        call("flm_replay_store", ptr(req_ready) if n1 else None, n1, ptr(req_switch) if n2 else None, n2, ptr(sel_row),
             ptr(obs), self.row_bytes, ptr(ro.net), ptr(ro.reward), ptr(ro.done), self.head, self.capacity,
             ptr(self.state), ptr(self.action), ptr(self.reward), ptr(self.done), ptr(self._slot), ptr(self._req),
             ptr(self._count), ro.stream)
        call("flm_observations_req_columns_rows", ptr(ro.cols), ptr(self._req), ptr(self._slot), n, ro.E, ro.A, b.Rp, b.T1,
             b.plane_stride, ptr(self.next_state), ro.stream)
        self._count_event.record(ro.b.stream)
        self._count_event.synchronize()
        k = int(self._count[0])
        self.head = (self.head + min(k, self.capacity)) % self.capacity
        self.size = min(self.capacity, self.size + k)
        return k

    def store_step_slots(self, ro, req_ready, req_switch, sel_row, obs, slots):
        """The feeding modes that keep an observation per agent between steps (crash penalty, src/main.py:271-273;
        switch-to-switch, src/main.py:280-296) and the default transitions beside them, for one rollout step of
        every environment: flm_replay_store_slots.  req_ready / req_switch / sel_row / obs are None when nobody
        consulted the network.  `slots` holds the per-agent state (curr_obs, has_curr, buf_obs, has_buf, buf_rew)
        and the two flags.  Returns the number of transitions (one host read-back)."""
        n1 = req_ready.numel() if req_ready is not None else 0
        n2 = req_switch.numel() if req_switch is not None else 0
        n = n1 + n2
        dev = self.device
        n_agents = ro.E * ro.A
        if getattr(self, "_slot", None) is None or self._slot.numel() < max(n, 1):
            cap = max(n, 1024)
            self._slot = torch.empty((cap,), dtype=torch.int32, device=dev)
            self._req = torch.empty((cap,), dtype=torch.int32, device=dev)
            # the kernels write the step's number of transitions into pinned host memory; the host reads it after an event
            self._count = torch.zeros((1,), dtype=torch.int32).pin_memory()
            self._count_event = torch.cuda.Event()
        need = query("flm_replay_store_slots_workspace", n_agents, n)
        if getattr(self, "_work", None) is None or self._work.numel() < need:
            self._work = torch.empty((max(need, 3 * (n_agents + 2048)),), dtype=torch.int32, device=dev)
        b = ro.b
        s2s = bool(slots.switch2switch)
        call("flm_replay_store_slots", ptr(req_ready) if n1 else None, n1, ptr(req_switch) if n2 else None, n2,
             ptr(sel_row) if n else None, ptr(obs) if n else None, self.row_bytes, ptr(ro.net), ptr(ro.reward), ptr(ro.done),
             ptr(ro.crash), n_agents, 1 if slots.crash_penalty else 0, 1 if s2s else 0, ptr(slots.curr_obs),
             ptr(slots.has_curr), ptr(slots.buf_obs) if s2s else None, ptr(slots.has_buf) if s2s else None,
             ptr(slots.buf_rew) if s2s else None, self.head, self.capacity, ptr(self.state), ptr(self.next_state),
             ptr(self.action), ptr(self.reward), ptr(self.done), ptr(self._work), ptr(self._slot), ptr(self._req),
             ptr(self._count), ro.stream)
        if n and not s2s:
            call("flm_observations_req_columns_rows", ptr(ro.cols), ptr(self._req), ptr(self._slot), n, ro.E, ro.A, b.Rp,
                 b.T1, b.plane_stride, ptr(self.next_state), ro.stream)
        self._count_event.record(ro.b.stream)
        self._count_event.synchronize()
        k = int(self._count[0])
        self.head = (self.head + min(k, self.capacity)) % self.capacity
        self.size = min(self.capacity, self.size + k)
        return k

    def sample_indices(self, batch_size):
        """batch_size distinct entries, uniformly (src/agent.py:161)."""
        if batch_size > self.size:
            raise ValueError("Sample larger than population")       # what random.sample raises
        idx = torch.empty((batch_size,), dtype=torch.int32, device=self.device)
        call("flm_replay_sample", self.size, int(batch_size), self.seed, self.draws & 0xFFFFFFFF, ptr(idx),
             torch.cuda.current_stream(self.device).cuda_stream)
        self.draws += 1
        return idx

    def sample(self, batch_size, used_bytes=0):
        """-> (states, actions, rewards, next_states, dones) of batch_size transitions.  The tensors are buffers of
        this object: the next call overwrites them.  used_bytes > 0 (rows of planes, row_shape = (planes, plane bytes)):
        only the first used_bytes of every plane are gathered, what lies behind them in the returned rows is stale."""
        B = int(batch_size)
        if self._out is None or self._out[0].shape[0] != B:
            dev = self.device
            # states and next states in one buffer, the next states right behind the states: the learner's
            # convolution then takes both in one launch
            both = torch.empty((2 * B,) + self.row_shape, dtype=self.dtype, device=dev)
            self._out = (both[:B], torch.empty((B,), dtype=torch.int64, device=dev),
                         torch.empty((B,), dtype=torch.float32, device=dev), both[B:],
                         torch.empty((B,), dtype=torch.float32, device=dev),
                         torch.empty((B,), dtype=torch.int32, device=dev))
        s, a, r, sn, d, idx = self._out
        if B > self.size:
            raise ValueError("Sample larger than population")
        st = torch.cuda.current_stream(self.device).cuda_stream
        # the draw and the gather are one launch
        used = (int(used_bytes) + 15) // 16 * 16
        planes = used and len(self.row_shape) == 2 and self.dtype == torch.uint8 and used < self.row_shape[1] and \
            self.row_shape[1] % 16 == 0
        shape = (self.row_shape[0], self.row_shape[1], used) if planes else (1, self.row_bytes, self.row_bytes)
        call("flm_replay_draw_gather", ptr(self.state), ptr(self.next_state), ptr(self.action), ptr(self.reward),
             ptr(self.done), self.size, self.seed, self.draws & 0xFFFFFFFF, B, self.row_bytes, *shape, ptr(s), ptr(sn),
             ptr(a), ptr(r), ptr(d), ptr(idx), st)
        self.draws += 1
        return s, a, r, sn, d
