"""Development check: the two distance methods agree on every valid state of every used row."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import flatland_marl_amd as fm
from flatland_marl_amd.synthetic import make_batch
names, E, A, depth, _ = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
E = min(E, 512)
data = make_batch(bench.load_fixtures(names), E, A, seed=0)
b = fm.ObservationBatch(data["grid"], data["start"], data["direction"], data["target"], data["speed"], depth)
b.reset()
d0 = b.dist.clone()
b.dist.fill_(12345)
b.compute_distances(method="bfs")
torch.cuda.synchronize()
d1 = b.dist
S = d0.shape[2]
valid_state = torch.arange(S, device=d0.device)[None, None, :] < (b.n_cells * 4)[:, None, None]
valid_row = torch.arange(A, device=d0.device)[None, :, None] < b.n_slots[:, None, None]
v = valid_state & valid_row
print("equal on valid:", bool((d0[v] == d1[v]).all()), "differ anywhere:", int((d0 != d1).sum()), "differ valid:", int(((d0 != d1) & v).sum()))
