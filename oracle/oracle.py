"""ctypes front end of the CPU checker (oracle/flmarl_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg; never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ERR_NAMES = {0: "", -1: "Capacity", -2: "IndexError", -3: "KeyError", -4: "AssertionError", -5: "RecursionError"}


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "libflmarl_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_shortest_path.restype = C.c_int
        _LIB.orc_altpaths.restype = C.c_int
        _LIB.orc_graph_build.restype = C.c_int
        _LIB.orc_bitmap_from_cells.restype = C.c_int
        _LIB.orc_get_obs.restype = C.c_int
        _LIB.orc_reorder_rails.restype = C.c_int
        _LIB.orc_check_crash.restype = C.c_int
        _LIB.orc_check_headon_crash.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Graph:
    """Output of rail_observations.py:410-510 for one grid."""

    def __init__(self, grid):
        grid = np.ascontiguousarray(grid, dtype=np.uint16)
        H, W = grid.shape
        self.grid, self.H, self.W = grid, H, W
        capN = H * W
        capR = 2 * H * W + 4
        capC = 8 * H * W + 16
        self.node_id = np.empty(H * W, np.int32)
        node_cell = np.empty(capN, np.int32)
        edge_info = np.empty(capR * 5, np.int32)
        off = np.empty(capR + 1, np.int32)
        cells = np.empty(capC, np.int32)
        self.cell_rail = np.empty(H * W, np.int32)
        counts = np.zeros(3, np.int32)
        rc = lib().orc_graph_build(_p(grid), H, W, _p(self.node_id), _p(node_cell), capN, _p(edge_info), capR,
                                   _p(off), _p(cells), capC, _p(self.cell_rail), _p(counts))
        if rc != 0:
            raise RuntimeError("orc_graph_build: %d" % rc)
        self.N, self.R, nc = int(counts[0]), int(counts[1]), int(counts[2])
        self.node_cell = node_cell[:self.N].copy()
        self.edge_info = edge_info[:self.R * 5].reshape(self.R, 5).copy()
        self.edge_cell_off = off[:self.R + 1].copy()
        self.edge_cells = cells[:nc].copy()


def distance_map(grid, tr, tc):
    grid = np.ascontiguousarray(grid, dtype=np.uint16)
    H, W = grid.shape
    d = np.empty(H * W * 4, np.int32)
    lib().orc_distance_map(_p(grid), H, W, int(tr), int(tc), _p(d))
    return d.reshape(H, W, 4)


def shortest_path(grid, dist, r, c, h, tr, tc, max_depth):
    H, W = grid.shape
    cap = max_depth + 2
    rows = np.empty(cap * 7, np.int32)
    n = lib().orc_shortest_path(_p(grid), H, W, _p(dist), int(r), int(c), int(h), int(tr), int(tc),
                                int(max_depth), _p(rows), cap)
    if n < -1:
        raise RuntimeError("orc_shortest_path capacity")
    return None if n < 0 else rows[:n * 7].reshape(n, 7).copy()


def altpaths(grid, dist, node_id, r, c, h, tr, tc, max_depth=500, cap_alt=32):
    """-> (list of [n,7] arrays, error name)"""
    H, W = grid.shape
    cap_len = max_depth + 80
    paths = np.empty(cap_alt * cap_len * 7, np.int32)
    lens = np.zeros(cap_alt, np.int32)
    n = lib().orc_altpaths(_p(grid), H, W, _p(dist), _p(node_id), int(r), int(c), int(h), int(tr), int(tc),
                           int(max_depth), _p(paths), _p(lens), cap_alt, cap_len)
    if n < 0:
        return [], ERR_NAMES[n]
    pv = paths.reshape(cap_alt, cap_len, 7)
    return [pv[i, :lens[i]].copy() for i in range(n)], ""


def prediction_cells(rows, sr, sc, tr, tc, tpc, depth):
    cells = np.empty((depth + 1) * 2, np.int32)
    if rows is None:
        rows_, n = np.zeros(7, np.int32), -1
    else:
        rows_, n = np.ascontiguousarray(rows, np.int32), rows.shape[0]
    lib().orc_prediction_cells(_p(rows_), n, int(sr), int(sc), int(tr), int(tc), int(tpc), int(depth), _p(cells))
    return cells.reshape(depth + 1, 2)


def bitmap_from_cells(g, cells, tr, tc, T1):
    cells = np.ascontiguousarray(cells, np.int32)
    bm = np.empty((g.R, T1), np.int8)
    rc = lib().orc_bitmap_from_cells(_p(cells), cells.shape[0], int(tr), int(tc), _p(g.node_id), _p(g.cell_rail),
                                     _p(g.edge_info), g.W, g.R, T1, _p(bm))
    return bm, ERR_NAMES[rc]


def shift_right(bm):
    lib().orc_shift_right(_p(bm), bm.shape[0], bm.shape[1])
    return bm


def get_obs(maps, handle, bitmap, Rp, reorder=False):
    A, R, T1 = maps.shape
    maps = np.ascontiguousarray(maps, np.int8)
    bitmap = np.ascontiguousarray(bitmap, np.int8)
    obs = np.empty((3 * Rp, T1), np.int16)
    rc = lib().orc_get_obs(_p(maps), A, R, T1, int(handle), _p(bitmap), int(Rp), _p(obs))
    if rc != 0:
        raise ValueError("could not broadcast: R > max_rails")
    if reorder:
        rc = lib().orc_reorder_rails(_p(obs), R, int(Rp), T1)
        if rc != 0:
            raise RuntimeError("reference does not terminate for this bitmap")
    return obs


def tpc_of(speed):
    """rail_observations.py:75-76 / predictions.py:99-100: int(np.reciprocal(speed))."""
    return int(np.reciprocal(np.float64(speed)))


class OracleEnv:
    """One environment through the whole observation pipeline (all agents READY_TO_DEPART)."""

    def __init__(self, fx, depth, num_agents=None):
        self.grid = np.ascontiguousarray(fx["grid"], np.uint16)
        self.H, self.W = self.grid.shape
        self.A = len(fx["handle"]) if num_agents is None else num_agents
        self.depth, self.T1 = depth, depth + 1
        self.start = np.asarray(fx["initial_position"][:self.A], np.int32)
        self.dir = np.asarray(fx["direction"][:self.A], np.int32)
        self.target = np.asarray(fx["target"][:self.A], np.int32)
        self.tpc = np.array([tpc_of(s) for s in fx["speed"][:self.A]], np.int32)
        self.g = Graph(self.grid)
        self._dist = {}

    def dist(self, a):
        t = (int(self.target[a, 0]), int(self.target[a, 1]))
        if t not in self._dist:
            self._dist[t] = distance_map(self.grid, *t)
        return self._dist[t]

    def path(self, a, max_depth=None):
        return shortest_path(self.grid, self.dist(a), self.start[a, 0], self.start[a, 1], self.dir[a],
                             self.target[a, 0], self.target[a, 1], self.T1 if max_depth is None else max_depth)

    def raw_bitmap(self, a, rows, sr=None, sc=None):
        sr = self.start[a, 0] if sr is None else sr
        sc = self.start[a, 1] if sc is None else sc
        cells = prediction_cells(rows, sr, sc, self.target[a, 0], self.target[a, 1], self.tpc[a], self.depth)
        return bitmap_from_cells(self.g, cells, self.target[a, 0], self.target[a, 1], self.T1)

    def get_many(self):
        """rail_observations.py:78-95 -> (maps int8 [A,R,T1], paths, errors per agent)."""
        maps = np.zeros((self.A, self.g.R, self.T1), np.int8)
        paths, errs = [], []
        for a in range(self.A):
            rows = self.path(a)
            paths.append(rows)
            bm, err = self.raw_bitmap(a, rows)
            errs.append(err)
            if not err:
                maps[a] = shift_right(bm)
        return maps, paths, errs

    def altmaps(self, a, r, c, h, ready=False):
        """rail_observations.py:97-111 for an agent standing at (r,c) heading h."""
        paths, err = altpaths(self.grid, self.dist(a), self.g.node_id, r, c, h, self.target[a, 0], self.target[a, 1])
        if err:
            return None, None, err
        out = []
        for rows in paths:
            bm, e = self.raw_bitmap(a, rows, r, c)
            if e:
                return None, None, e
            if ready:
                bm[:, -1] = 0
                shift_right(bm)
            out.append(bm)
        return out, paths, ""
