"""Device-resident observation state for a batch of E independent environments.

All environments of a batch share the grid size H x W, the agent count A and the
prediction depth.  Everything lives in HBM as torch tensors; the kernels are the
C ABI of include/flmarl.h.  E = 1 reproduces the single-environment objects of
the reference (RailObsForRailEnv / ObsPreprocessor, see rail_observations.py and
preprocessing.py in this package for the drop-in classes).

Layout in HBM (E environments, A agents, Rp padded rails, T1 = depth + 1):
  grid      [E,H,W]  uint16      node_id/cell_rail/cell_idx [E,H*W] int32
  edge_info [E,capR,5] int32     pred [E,capCells*4,4] uint16
  dist      [E,A,capCells*4] uint16 (one row per distinct target, tslot maps agents to rows)
  path_cell/path_meta [E,A,Lcap]  int32 / uint32
  maps      [E,A,plane_stride] int8,   plane = [Rp,T1] rows, plane_stride = Rp*T1 rounded up to 16
  obs       [E,A,3,plane_stride] uint8 (own plane, positive heat, negative heat)
"""
import numpy as np
import torch

from . import _lib
from ._lib import call, ptr

READY_TO_DEPART, ACTIVE, DONE, DONE_REMOVED = 0, 1, 2, 3


def times_per_cell(speed):
    """int(np.reciprocal(speed)) as in the reference (src/rail_observations.py:75-76)."""
    return np.reciprocal(np.asarray(speed, dtype=np.float64)).astype(np.int64).astype(np.int32)


def _dev_i32(x, device):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.int32), device=device)


def plane_stride_for(Rp, T1):
    """Bytes between consecutive planes: Rp*T1 rounded up to 256, so that every plane starts on a cache-line
    boundary and whole-wave 1 KiB stores never straddle one (the C ABI itself only asks for a multiple of 16)."""
    return (Rp * T1 + 255) // 256 * 256


class ObservationBatch:
    def __init__(self, grid, start_rc, direction, target_rc, speed, depth, max_rails=None, status=None,
                 device="cuda:0", stream=None):
        if not torch.cuda.is_available():
            raise _lib.FlmError("no GPU visible: the observation path has no CPU implementation")
        _lib.load()
        self.device = torch.device(device)
        grid = np.ascontiguousarray(grid, dtype=np.uint16)
        if grid.ndim == 2:
            grid = grid[None]
        self.E, self.H, self.W = grid.shape
        start_rc = np.asarray(start_rc).reshape(self.E, -1, 2)
        self.A = start_rc.shape[1]
        target_rc = np.asarray(target_rc).reshape(self.E, self.A, 2)
        self.depth, self.T1 = int(depth), int(depth) + 1
        self.max_rails = max_rails
        # uint16 has no arithmetic in torch; the kernels only need the bytes
        self.grid = torch.as_tensor(grid.view(np.int16), device=self.device)
        self.start_cell = _dev_i32(start_rc[..., 0] * self.W + start_rc[..., 1], self.device)
        self.pos_cell = self.start_cell.clone()
        self.direction = _dev_i32(np.asarray(direction).reshape(self.E, self.A), self.device)
        self.target_cell = _dev_i32(target_rc[..., 0] * self.W + target_rc[..., 1], self.device)
        self.tpc = _dev_i32(times_per_cell(speed).reshape(self.E, self.A), self.device)
        st = np.zeros((self.E, self.A), np.int32) if status is None else np.asarray(status).reshape(self.E, self.A)
        self.status = _dev_i32(st, self.device)
        self.cap_cells = int((grid != 0).reshape(self.E, -1).sum(1).max())
        if self.cap_cells * 4 >= 65535:
            raise ValueError("more than 16383 rail cells per environment are not supported")
        self.cap_cells = max(self.cap_cells + (self.cap_cells & 1), 2)   # even: distance rows stay 16-byte aligned
        self.R = None
        self.maps = None
        self.obs = None
        # every kernel of this batch is enqueued on one stream (default: the stream current at construction)
        self.stream = torch.cuda.current_stream(self.device) if stream is None else stream

    # ------------------------------------------------------------------ graph
    def _alloc_graph(self, capN, capR, capC):
        E, HW, dev = self.E, self.H * self.W, self.device
        i32 = dict(dtype=torch.int32, device=dev)
        self.capN, self.capR, self.capC = capN, capR, capC
        self.node_id = torch.empty((E, HW), **i32)
        self.node_cell = torch.empty((E, capN), **i32)
        self.edge_info = torch.empty((E, capR, 5), **i32)
        self.edge_cell_off = torch.empty((E, capR + 1), **i32)
        self.edge_cells = torch.empty((E, capC), **i32)
        self.cell_rail = torch.empty((E, HW), **i32)
        self.counts = torch.zeros((E, 4), **i32)
        self.cell_idx = torch.empty((E, HW), **i32)
        self.n_cells = torch.empty((E,), **i32)
        self.idx_cell = torch.empty((E, self.cap_cells), **i32)
        self.pred = None     # reverse-move table: only the breadth-first fallback needs it

    def _launch_graph(self):
        E, H, W = self.E, self.H, self.W
        self._chain_ready = False                            # a chain index built before this launch is stale
        s = self.stream.cuda_stream
        call("flm_graph_build", ptr(self.grid), E, H, W, self.capN, self.capR, self.capC, ptr(self.node_id),
             ptr(self.node_cell), ptr(self.edge_info), ptr(self.edge_cell_off), ptr(self.edge_cells),
             ptr(self.cell_rail), ptr(self.counts), s)
        call("flm_state_index", ptr(self.grid), E, H, W, self.cap_cells, ptr(self.cell_idx), ptr(self.n_cells),
             None, ptr(self.counts), ptr(self.idx_cell), s)

    def build_graph(self, sync=True):
        """Switch graph + cell->rail table + state numbering for the BFS.

        sync=True reads the per-environment counts back (sizes the rail axis, grows capacities, raises on
        overflow).  sync=False only enqueues the kernels into buffers sized by an earlier synchronous call;
        check_graph() validates afterwards."""
        if not sync:
            if self.R is None:
                raise _lib.FlmError("build_graph(sync=False) needs one synchronous call first")
            self._launch_graph()
            return self
        if getattr(self, "capN", None) is None:
            capN = min(_lib.MAX_NODES, self.cap_cells)
            self._alloc_graph(capN, 2 * capN + 8, 2 * self.cap_cells + 2 * (self.H + self.W))
        for attempt in range(3):
            self._launch_graph()
            counts = self.counts.cpu().numpy()
            errs = int(np.bitwise_or.reduce(counts[:, 3]))
            if errs == 0:
                break
            if errs & _lib.GRAPH_ERR_NODES:
                raise ValueError("more than %d switches in one environment" % _lib.MAX_NODES)
            self._alloc_graph(self.capN, max(self.capR, int(counts[:, 1].max())), max(self.capC, int(counts[:, 2].max())))
        else:
            raise _lib.FlmError("graph build did not converge")
        if not getattr(self, "_tight", False):
            # shrink the capacities to what the grids need (plus a margin): the port tables of the graph
            # kernel live in LDS, and a small node capacity lets it stage the grid there as well
            self._tight = True
            capN = int(counts[:, 0].max()) + 16
            if capN < self.capN:
                self._alloc_graph(capN, int(counts[:, 1].max()) + 16, int(counts[:, 2].max()) + 64)
                self._launch_graph()
                counts = self.counts.cpu().numpy()
                assert int(np.bitwise_or.reduce(counts[:, 3])) == 0
        self.num_nodes = counts[:, 0].copy()
        self.R = counts[:, 1].copy()
        need = int(self.R.max())
        if self.max_rails is not None and self.max_rails < need:
            # the reference fails in _fill_padding (src/preprocessing.py:19) with a numpy broadcast error
            raise ValueError("could not broadcast: an environment has %d rails > max_rails=%d" % (need, self.max_rails))
        self.Rp = max(need, 1) if self.max_rails is None else int(self.max_rails)
        self.plane_stride = plane_stride_for(self.Rp, self.T1)
        return self

    def check_graph(self):
        """Raise if any asynchronous build_graph since the last check overflowed or changed the rail counts."""
        counts = self.counts.cpu().numpy()
        flags = int(np.bitwise_or.reduce(counts[:, 3]))
        if flags & _lib.GRAPH_ERR_STATES:
            self.lds_anchor = self.lds_slots = None          # outgrown: sized again by the next chain_index()
        if flags != 0 or int(counts[:, 1].max()) > self.Rp:
            raise _lib.FlmError("graph capacities exceeded in an asynchronous build")

    # ------------------------------------------------------------------ distances and paths
    def compute_distances(self, method=None):
        """Distance of every (rail cell, heading) state to every distinct target of every environment.

        method "chains" (default when the tables fit in LDS) or "bfs" (wave-per-target breadth-first search)."""
        E, A, dev = self.E, self.A, self.device
        i32 = dict(dtype=torch.int32, device=dev)
        s = self.stream.cuda_stream
        capS = self.cap_cells * 4
        if getattr(self, "tslot", None) is None:
            self.tslot = torch.empty((E, A), **i32)
            self.n_slots = torch.empty((E,), **i32)
            self.slot_target = torch.empty((E, A), **i32)
        if getattr(self, "dist", None) is None:
            self.dist = torch.empty((E, A, capS), dtype=torch.int16, device=dev)
        if method is None:
            method = "chains" if capS * 6 + capS // 8 + 1024 <= 159 * 1024 else "bfs"
        if method == "bfs":
            self._succ_valid = False
            if self.pred is None:
                self.pred = torch.empty((E, capS, 4), dtype=torch.int16, device=dev)
            call("flm_state_index", ptr(self.grid), E, self.H, self.W, self.cap_cells, ptr(self.cell_idx),
                 ptr(self.n_cells), ptr(self.pred), ptr(self.counts), ptr(self.idx_cell), s)
            call("flm_distance_bfs", ptr(self.pred), ptr(self.cell_idx), ptr(self.n_cells), ptr(self.target_cell), E,
                 A, self.H, self.W, self.cap_cells, ptr(self.tslot), ptr(self.n_slots), ptr(self.slot_target),
                 ptr(self.dist), s)
            return self
        if not getattr(self, "_chain_ready", False):         # reset_fused() may have built it for these grids already
            self.chain_index()
        self._chain_ready = False                            # one use: positions and targets may change before the next call
        if self.lds_anchor * 80 > 159 * 1024:                # the anchors' labels of 32 targets do not fit the LDS
            return self.compute_distances(method="bfs")
        call("flm_distance_chains", ptr(self.chain_root), ptr(self.chain_steps), ptr(self.chain_edges),
             ptr(self.n_anchor), ptr(self.n_cells), ptr(self.cell_idx), ptr(self.slot_target), ptr(self.n_slots), E, A,
             self.H, self.W, self.cap_cells, self.cap_anchor, self.lds_anchor, ptr(self.dist), ptr(self.counts), s)
        return self

    def chain_index(self):
        """Distinct targets per environment and the chain index of every grid (anchors, forced-move chains, the
        walk tables): what both the dense distance maps and the fused path kernel start from."""
        E, A, dev = self.E, self.A, self.device
        i32 = dict(dtype=torch.int32, device=dev)
        s = self.stream.cuda_stream
        capS = self.cap_cells * 4
        if getattr(self, "tslot", None) is None:
            self.tslot = torch.empty((E, A), **i32)
            self.n_slots = torch.empty((E,), **i32)
            self.slot_target = torch.empty((E, A), **i32)
        if getattr(self, "chain_root", None) is None:
            # anchors: states with a choice (at most two per switch cell and heading pair), target-cell states,
            # and the few states with one move that a walk does not treat as forced (dead ends)
            self.cap_anchor = int(min(0x7FF0, 4 * self.capN + 4 * A + 256))
            self.chain_root = torch.empty((E, capS), dtype=torch.int16, device=dev)
            self.chain_steps = torch.empty((E, capS), dtype=torch.int16, device=dev)
            self.chain_edges = torch.empty((E, self.cap_anchor, 4), **i32)
            self.n_anchor = torch.empty((E,), **i32)
            self.chain_succ = torch.empty((E, capS), dtype=torch.int16, device=dev)
            self.chain_choice = torch.empty((E, capS, 4), dtype=torch.int16, device=dev)
            self.chain_walk = self.move_edge = self.move_succ = None
            if capS <= 32768:
                self.chain_walk = torch.empty((E, capS), dtype=torch.int16, device=dev)
                self.move_edge = torch.empty((E, self.cap_anchor, 4), **i32)
                self.move_succ = torch.empty((E, self.cap_anchor, 4), dtype=torch.int16, device=dev)
        call("flm_target_slots", ptr(self.target_cell), E, A, ptr(self.tslot), ptr(self.n_slots), ptr(self.slot_target), s)
        call("flm_chain_index", ptr(self.grid), ptr(self.cell_idx), ptr(self.n_cells), ptr(self.target_cell), E, A,
             self.H, self.W, self.cap_cells, self.cap_anchor, ptr(self.chain_root), ptr(self.chain_steps),
             ptr(self.chain_edges), ptr(self.n_anchor), ptr(self.counts), ptr(self.chain_succ), ptr(self.chain_choice),
             ptr(self.chain_walk), ptr(self.move_edge), ptr(self.move_succ), ptr(self.idx_cell), s)
        self._succ_valid = True
        self._chain_ready = True                             # the chain index of the current grids and targets is in place
        if getattr(self, "lds_anchor", None) is None:
            # anchors = states with a choice of moves (fixed by the grid) + states on target cells (<= 4*A).
            # The LDS tables are sized for the count seen now plus a margin for later target assignments;
            # an environment that outgrows it is flagged in counts: check_graph() then forgets this size, and
            # the next chain_index() takes it from the counts of that launch.
            n_now = int(self.n_anchor.max().item())
            self.lds_anchor = int(min(self.cap_anchor, n_now + min(4 * A, 64)))
            # distinct targets per environment, rounded up to the eight targets one decision word holds
            self.lds_slots = (int(self.n_slots.max().item()) + 7) // 8 * 8
        return self

    def fused_fits(self, max_depth=None):
        """Can paths_columns_fused run for this batch (tables within the LDS and the kernel's index widths)?"""
        from ._lib import query
        if getattr(self, "chain_walk", None) is None or self.H > 256 or self.W > 256 or self.T1 > 512:
            return False
        Lcap = (self.T1 if max_depth is None else int(max_depth)) + 1
        need = query("flm_paths_columns_lds_bytes", self.A, self.cap_cells, self.lds_anchor, self.lds_slots, self.capR, Lcap,
                     self.T1)
        return 0 < need <= 160 * 1024 - 512

    def paths_columns_fused(self, shift=1, store_paths=True, max_depth=None):
        """shortest_paths() + columns() in one launch straight from the chain index (no dense distance maps)."""
        E, A, dev = self.E, self.A, self.device
        max_depth = self.T1 if max_depth is None else int(max_depth)
        Lcap = max_depth + 1
        if store_paths and getattr(self, "Lcap", None) != Lcap:
            self.path_cell = torch.empty((E, A, Lcap), dtype=torch.int32, device=dev)
            self.path_meta = torch.empty((E, A, Lcap), dtype=torch.int32, device=dev)
            self.Lcap = Lcap
        if getattr(self, "path_len", None) is None:
            self.path_len = torch.empty((E, A), dtype=torch.int32, device=dev)
        if getattr(self, "cols", None) is None:
            self.cols = torch.empty((E, A, self.T1), dtype=torch.int16, device=dev)
        if getattr(self, "agent_err", None) is None:
            self.agent_err = torch.empty((E, A), dtype=torch.int32, device=dev)
        from ._lib import query
        n_ws = query("flm_paths_columns_workspace", E, A, self.lds_anchor, self.lds_slots)
        if getattr(self, "_fused_ws", None) is None or self._fused_ws.numel() < n_ws:
            self._fused_ws = torch.empty((n_ws,), dtype=torch.int32, device=dev)
        call("flm_paths_columns", ptr(self.chain_walk), ptr(self.chain_root), ptr(self.chain_steps), ptr(self.chain_edges),
             ptr(self.move_edge),
             ptr(self.move_succ), ptr(self.n_anchor), ptr(self.n_cells), ptr(self.cell_idx), ptr(self.idx_cell), ptr(self.cell_rail),
             ptr(self.node_id), ptr(self.edge_info), ptr(self.tslot), ptr(self.n_slots), ptr(self.slot_target),
             ptr(self.pos_cell), ptr(self.direction), ptr(self.pos_cell), ptr(self.target_cell), ptr(self.status),
             ptr(self.tpc), E, A, self.H, self.W, self.cap_cells, self.cap_anchor, self.lds_anchor, self.lds_slots,
             self.capR, self.Rp, self.T1, max_depth, Lcap, int(shift), ptr(self._fused_ws),
             ptr(self.path_cell) if store_paths else None, ptr(self.path_meta) if store_paths else None,
             ptr(self.path_len), ptr(self.cols), ptr(self.agent_err), ptr(self.counts), self.stream.cuda_stream)
        return self.cols

    def succ_table(self):
        """Forced-move table of the chain index (None after a breadth-first distance computation)."""
        return self.chain_succ if getattr(self, "_succ_valid", False) else None

    def choice_table(self):
        """Per state without a forced successor the states its moves lead to (written with the successor table), or
        None."""
        return self.chain_choice if getattr(self, "_succ_valid", False) else None

    def shortest_paths(self, max_depth=None, start_cell=None, start_dir=None):
        """get_shortest_paths for every agent (src/utils/shortest_path.py:11-90)."""
        E, A, dev = self.E, self.A, self.device
        max_depth = self.T1 if max_depth is None else int(max_depth)
        Lcap = max_depth + 1
        if getattr(self, "Lcap", None) == Lcap:
            path_cell, path_meta, path_len = self.path_cell, self.path_meta, self.path_len
        else:
            path_cell = torch.empty((E, A, Lcap), dtype=torch.int32, device=dev)
            path_meta = torch.empty((E, A, Lcap), dtype=torch.int32, device=dev)
            path_len = torch.empty((E, A), dtype=torch.int32, device=dev)
        sc = self.pos_cell if start_cell is None else start_cell
        sd = self.direction if start_dir is None else start_dir
        call("flm_shortest_paths", ptr(self.grid), ptr(self.cell_idx), ptr(self.dist), ptr(self.succ_table()),
             ptr(self.choice_table()), ptr(self.tslot), ptr(sc),
             ptr(sd), ptr(self.target_cell), ptr(self.status), E, A, self.H, self.W, self.cap_cells, max_depth, Lcap,
             ptr(path_cell), ptr(path_meta), ptr(path_len), self.stream.cuda_stream)
        self.path_cell, self.path_meta, self.path_len, self.Lcap = path_cell, path_meta, path_len, Lcap
        return self

    # ------------------------------------------------------------------ bitmaps and observations
    def bitmaps(self, shift=1, out=None):
        """Occupancy planes of the current paths; shift=1 is get_many (src/rail_observations.py:78-95)."""
        E, A, dev = self.E, self.A, self.device
        if out is None:
            shape = (E, A, self.plane_stride)
            out = self.maps if (self.maps is not None and tuple(self.maps.shape) == shape) else \
                torch.empty(shape, dtype=torch.int8, device=dev)
        if getattr(self, "agent_err", None) is None:
            self.agent_err = torch.empty((E, A), dtype=torch.int32, device=dev)
        if getattr(self, "cols", None) is None:
            self.cols = torch.empty((E, A, self.T1), dtype=torch.int16, device=dev)
        call("flm_bitmaps", ptr(self.path_cell), ptr(self.path_len), None, ptr(self.pos_cell), ptr(self.target_cell),
             ptr(self.tpc), ptr(self.node_id), ptr(self.cell_rail), ptr(self.edge_info), E, A, E * A, self.H, self.W,
             self.capR, self.Rp, self.T1, self.plane_stride, self.Lcap, int(shift), ptr(self.cols), ptr(out),
             ptr(self.agent_err), self.stream.cuda_stream)
        self.maps = out
        return out

    def columns(self, shift=1):
        """Column form of the occupancy planes of the current paths: [E,A,T1] int16, rail*2 + (sign > 0) or -1."""
        E, A, dev = self.E, self.A, self.device
        if getattr(self, "cols", None) is None:
            self.cols = torch.empty((E, A, self.T1), dtype=torch.int16, device=dev)
        if getattr(self, "agent_err", None) is None:
            self.agent_err = torch.empty((E, A), dtype=torch.int32, device=dev)
        call("flm_columns", ptr(self.path_cell), ptr(self.path_len), None, ptr(self.pos_cell), ptr(self.target_cell),
             ptr(self.tpc), ptr(self.node_id), ptr(self.cell_rail), ptr(self.edge_info), E, A, E * A, self.H, self.W,
             self.capR, self.Rp, self.T1, self.Lcap, int(shift), ptr(self.cols), ptr(self.agent_err),
             self.stream.cuda_stream)
        return self.cols

    def observations_fused(self, out=None):
        """Observations of every agent straight from the column form; obs[:, :, 0] are the dense planes."""
        if self.A > 255:
            raise ValueError("heat counts are stored in 8 bits: at most 255 agents per environment")
        if out is None:
            shape = (self.E, self.A, 3, self.plane_stride)
            out = self.obs if (self.obs is not None and tuple(self.obs.shape) == shape) else \
                torch.empty(shape, dtype=torch.uint8, device=self.device)
        call("flm_observations_columns", ptr(self.cols), self.E, self.A, self.Rp, self.T1, self.plane_stride, ptr(out),
             self.stream.cuda_stream)
        self.obs = out
        return out

    def reset_fused(self, sync=True, fused_paths=None, store_paths=True):
        """reset() + observations() without the dense intermediate planes.  fused_paths (default: whenever the
        tables fit the LDS) computes paths and columns straight from the chain index, without the dense distance
        maps; False takes distance maps -> shortest paths -> columns."""
        if self.T1 > 512:
            # the column kernels hold a path's time steps in registers (eight per lane): a longer horizon goes
            # through the dense planes, with the same observations
            if fused_paths:
                raise _lib.FlmError("the fused path kernel holds at most 512 time steps")
            self.reset(sync=sync)
            return self.observations()
        self.build_graph(sync=sync)
        capS = self.cap_cells * 4
        chains = capS * 6 + capS // 8 + 1024 <= 159 * 1024
        if fused_paths is not False and chains:
            self.chain_index()
            if self.fused_fits():
                self.paths_columns_fused(shift=1, store_paths=store_paths)
                return self.observations_fused()
            if fused_paths:
                raise _lib.FlmError("the fused path kernel's tables do not fit the LDS for this batch")
        self.compute_distances()
        self.shortest_paths()
        self.columns(shift=1)
        return self.observations_fused()

    def observations(self, maps=None, own=None, out=None):
        """[E,A,3,plane_stride] uint8: own plane, positive heat, negative heat (src/preprocessing.py:24-30,61-74)."""
        maps = self.maps if maps is None else maps
        if self.A > 255:
            raise ValueError("heat counts are stored in 8 bits: at most 255 agents per environment")
        if out is None:
            shape = (self.E, self.A, 3, self.plane_stride)
            out = self.obs if (self.obs is not None and tuple(self.obs.shape) == shape) else \
                torch.empty(shape, dtype=torch.uint8, device=self.device)
        call("flm_observations", ptr(maps), ptr(own), self.E, self.A, self.plane_stride, ptr(out),
             self.stream.cuda_stream)
        self.obs = out
        return out

    def reset(self, sync=True, out=None):
        """Graph, distances, shortest paths, occupancy planes: what env.reset() triggers in the reference."""
        self.build_graph(sync=sync)
        self.compute_distances()
        self.shortest_paths()
        return self.bitmaps(shift=1, out=out)

    # ------------------------------------------------------------------ rollout
    def set_agent_state(self, status=None, pos_cell=None, direction=None):
        """Upload the agents' current status / cell (-1 = off the grid) / heading, each [E,A]."""
        if status is not None:
            self.status.copy_(_dev_i32(np.asarray(status).reshape(self.E, self.A), self.device))
        if pos_cell is not None:
            self.pos_cell.copy_(_dev_i32(np.asarray(pos_cell).reshape(self.E, self.A), self.device))
        if direction is not None:
            self.direction.copy_(_dev_i32(np.asarray(direction).reshape(self.E, self.A), self.device))

    def altpaths(self, req_agent, max_depth=500, cap_alt=8):
        """get_altpaths (src/utils/shortest_path.py:92-163) for the agents e*A+a listed in req_agent."""
        req = _dev_i32(np.asarray(req_agent).reshape(-1), self.device)
        n = req.numel()
        Lcap = max_depth + 40
        dev = self.device
        out = dict(
            count=torch.empty((n,), dtype=torch.int32, device=dev),
            len=torch.zeros((n, cap_alt), dtype=torch.int32, device=dev),
            cell=torch.empty((n, cap_alt, Lcap), dtype=torch.int32, device=dev),
            meta=torch.empty((n, cap_alt, Lcap), dtype=torch.int32, device=dev),
            err=torch.empty((n,), dtype=torch.int32, device=dev), req=req, Lcap=Lcap, cap_alt=cap_alt)
        call("flm_altpaths", ptr(self.grid), ptr(self.cell_idx), ptr(self.dist), ptr(self.succ_table()),
             ptr(self.choice_table()), ptr(self.idx_cell), ptr(self.tslot), ptr(self.node_id),
             ptr(req), n, ptr(self.pos_cell), ptr(self.direction), ptr(self.target_cell), self.E, self.A, self.H,
             self.W, self.cap_cells, int(max_depth), cap_alt, Lcap, ptr(out["count"]), ptr(out["len"]),
             ptr(out["cell"]), ptr(out["meta"]), ptr(out["err"]), self.stream.cuda_stream)
        return out

    def planes_for_paths(self, path_cell, path_len, agent_flat, Lcap, shift):
        """Occupancy planes [n,plane_stride] of arbitrary paths (alternatives) of the agents agent_flat[n]."""
        n = agent_flat.numel()
        planes = torch.empty((n, self.plane_stride), dtype=torch.int8, device=self.device)
        err = torch.empty((n,), dtype=torch.int32, device=self.device)
        cols = torch.empty((n, self.T1), dtype=torch.int16, device=self.device)
        call("flm_bitmaps", ptr(path_cell), ptr(path_len), ptr(agent_flat), ptr(self.pos_cell), ptr(self.target_cell),
             ptr(self.tpc), ptr(self.node_id), ptr(self.cell_rail), ptr(self.edge_info), self.E, self.A, n, self.H,
             self.W, self.capR, self.Rp, self.T1, self.plane_stride, int(Lcap), int(shift), ptr(cols), ptr(planes),
             ptr(err), self.stream.cuda_stream)
        return planes, err

    def observations_for(self, maps, req_agent, own_planes=None):
        """[n,3,plane_stride] uint8 observations of selected agents (src/preprocessing.py:61-74)."""
        req = req_agent if torch.is_tensor(req_agent) else _dev_i32(np.asarray(req_agent).reshape(-1), self.device)
        n = req.numel()
        out = torch.empty((n, 3, self.plane_stride), dtype=torch.uint8, device=self.device)
        call("flm_observations_req", ptr(maps), ptr(req), ptr(own_planes), n, self.E, self.A, self.plane_stride,
             ptr(out), self.stream.cuda_stream)
        return out

    def reorder_rails(self, obs):
        """Rail reordering (src/preprocessing.py:32-59) of [...,3,plane_stride] uint8 observations
        -> (reordered copy, err [n] int32; 6 = the reference's loop would not end on that plane)."""
        flat = obs.reshape(-1, 3, self.plane_stride)
        n = flat.shape[0]
        out = torch.empty_like(flat)
        err = torch.zeros((n,), dtype=torch.int32, device=self.device)
        call("flm_reorder_rails", ptr(flat), ptr(out), n, self.Rp, self.T1, self.plane_stride, ptr(err),
             self.stream.cuda_stream)
        return out.view(obs.shape), err

    def check_crash(self, maps, agent_sel, before_switch, next_cell, init_cell=None):
        """check_crash (src/rail_observations.py:206-230) for one selected agent per environment -> [E] int32."""
        dev = self.device
        crash = torch.zeros((self.E,), dtype=torch.int32, device=dev)
        init = self.start_cell if init_cell is None else init_cell
        # the three selector tensors must stay referenced until the launch has been enqueued
        sel, bsw, nxt = _dev_i32(agent_sel, dev), _dev_i32(before_switch, dev), _dev_i32(next_cell, dev)
        call("flm_check_crash", ptr(maps), ptr(self.status), ptr(self.tpc), ptr(self.pos_cell), ptr(init), ptr(sel),
             ptr(bsw), ptr(nxt), self.E, self.A, self.Rp, self.T1, self.plane_stride, ptr(crash),
             self.stream.cuda_stream)
        return crash

    def update_bitmaps(self, maps, agent_sel, before_switch):
        """update_bitmaps (src/rail_observations.py:232-252) for one selected agent per environment, in place."""
        dev = self.device
        sel, bsw = _dev_i32(agent_sel, dev), _dev_i32(before_switch, dev)
        call("flm_update_bitmaps", ptr(maps), ptr(self.status), ptr(self.tpc), ptr(sel), ptr(bsw), self.E, self.A,
             self.Rp, self.T1, self.plane_stride, self.stream.cuda_stream)
        return maps

    # ------------------------------------------------------------------ host views (tests, drop-in classes)
    def upload_maps(self, host_maps, e=0, out=None):
        """[A,R,T1] host array of one environment -> device planes [E,A,plane_stride] (rows >= R zero)."""
        if out is None:
            out = torch.zeros((self.E, self.A, self.plane_stride), dtype=torch.int8, device=self.device)
        m = np.zeros((self.A, self.Rp, self.T1), np.int8)
        hm = np.asarray(host_maps)
        m[:, : hm.shape[1], :] = hm
        out[e, :, : self.Rp * self.T1] = torch.as_tensor(m.reshape(self.A, -1), device=self.device)
        return out

    def plane_numpy(self, plane, R=None):
        """One device plane [plane_stride] -> [R,T1] int8."""
        R = self.Rp if R is None else R
        return plane[: self.Rp * self.T1].reshape(self.Rp, self.T1)[:R].cpu().numpy()

    def maps_numpy(self, e=0, maps=None):
        """[A,R_e,T1] int8 view of one environment, as the reference's get_many returns it."""
        maps = self.maps if maps is None else maps
        m = maps[e, :, : self.Rp * self.T1].reshape(self.A, self.Rp, self.T1)
        return m[:, : int(self.R[e])].cpu().numpy()

    def obs_numpy(self, e=0, obs=None):
        """[A,3,Rp,T1] int16 of one environment (own plane sign-extended)."""
        obs = self.obs if obs is None else obs
        o = obs[e, :, :, : self.Rp * self.T1].reshape(self.A, 3, self.Rp, self.T1).cpu().numpy()
        out = o.astype(np.int16)
        out[:, 0] = o[:, 0].view(np.int8)
        return out

    def paths_numpy(self, e=0):
        """List over agents of [n,7] int32 rows (r, c, heading, action, next r, next c, next heading) or None."""
        W = self.W
        pc = self.path_cell[e].cpu().numpy()
        pm = self.path_meta[e].cpu().numpy().view(np.uint32)
        pl = self.path_len[e].cpu().numpy()
        dr = np.array([-1, 0, 1, 0])
        dc = np.array([0, 1, 0, -1])
        out = []
        for a in range(self.A):
            n = int(pl[a])
            if n < 0:
                out.append(None)
                continue
            cell, meta = pc[a, :n].astype(np.int64), pm[a, :n]
            h, act, nh = meta & 0xFF, (meta >> 8) & 0xFF, (meta >> 16) & 0xFF
            r, c = cell // W, cell % W
            move = act != 4
            rows = np.stack([r, c, h, act, r + np.where(move, dr[nh], 0), c + np.where(move, dc[nh], 0), nh], 1)
            out.append(rows.astype(np.int32))
        return out
