"""Where the time of the one-workgroup-per-environment kernels of the reset path goes (graph build, chain index, anchor
decisions, walks + columns): shader-clock stamps at their phase boundaries (diagnostic hook flm_debug_fused_stamps),
median over the environments of a cfg3-shaped batch.

    python tools/fused_phases.py [envs] [workload]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import flatland_marl_amd as fm  # noqa: E402
from flatland_marl_amd import _lib  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
data, E, A, depth, desc = bench.workload_batch(sys.argv[2] if len(sys.argv) > 2 else "cfg3", E)
b = fm.ObservationBatch(data["grid"], data["start"], data["direction"], data["target"], data["speed"], depth)
b.reset_fused()
torch.cuda.synchronize()
lib = _lib.load()
need = _lib.query("flm_paths_columns_lds_bytes", A, b.cap_cells, b.lds_anchor, b.lds_slots, b.capR, b.T1 + 1, b.T1)
print("cap_cells", b.cap_cells, "lds_anchor", b.lds_anchor, "lds_slots", b.lds_slots, "LDS bytes", need,
      "anchors max", int(b.n_anchor.max()), "slots max", int(b.n_slots.max()))
st = torch.zeros((3, E, 16), dtype=torch.int64, device=b.device)
lib.flm_debug_fused_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.flm_debug_fused_stamps.restype = None
lib.flm_debug_fused_stamps(st.data_ptr(), E)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
b.build_graph(sync=False)
b.chain_index()
ev0.record()
b.paths_columns_fused()
ev1.record()
torch.cuda.synchronize()
lib.flm_debug_fused_stamps(None, 0)
sall = st.cpu().numpy()
s = sall[0]
print("both kernels %.3f ms" % ev0.elapsed_time(ev1))
names = {(0, 1): "K1 load+init", (1, 2): "K1 sweeps", (2, 3): "K1 decisions", (4, 5): "K2 tables", (5, 6): "K2 hops",
         (6, 7): "K2 runs", (7, 8): "K2 paths out", (8, 9): "K2 columns"}
for (i, j), n in names.items():
    d = s[:, j] - s[:, i]
    print("%-14s median %8d cycles   max %8d" % (n, np.median(d), d.max()))
print("K1 residency median %d cycles, K2 %d" % (np.median(s[:, 3] - s[:, 0]), np.median(s[:, 9] - s[:, 4])))
print("rounds of group 0: median %d max %d" % (np.median(s[:, 15]), s[:, 15].max()))
for region, title, names in ((1, "graph_build", {(0, 1): "nodes", (1, 2): "port walks", (2, 3): "first port per far port",
                                                     (3, 4): "edge scan + emitting walks"}),
                             (2, "chain_index", {(0, 1): "classify", (1, 2): "succ out + pointer doubling",
                                                 (2, 3): "anchor numbering", (3, 4): "root/steps/walk out",
                                                 (4, 5): "edges"})):
    r = sall[region]
    last = max(j for _, j in names)
    print("%s residency median %d cycles" % (title, np.median(r[:, last] - r[:, 0])))
    for (i, j), n in names.items():
        d = r[:, j] - r[:, i]
        print("  %-28s median %8d cycles   max %8d" % (n, np.median(d), d.max()))
