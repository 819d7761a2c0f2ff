"""Per-kernel timings and variant sweeps on one GPU (development tool, not a test).

    python tools/kernel_bench.py [--workload cfg3] [--envs E]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import flatland_marl_amd as fm  # noqa: E402


def timed(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for a, b in ev]
    return float(np.median(t)), float(np.min(t))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--envs", type=int, default=0)
    ap.add_argument("--stages", default="graph,bfs,paths,bitmaps,obs,fused")
    args = ap.parse_args()
    data, E, A, depth, desc = bench.workload_batch(args.workload, args.envs)
    b = fm.ObservationBatch(data["grid"], data["start"], data["direction"], data["target"], data["speed"], depth)
    b.reset()
    b.observations()
    ref_maps = b.maps.clone()
    ref_obs_sum = int(b.obs.view(torch.int64).sum().item())

    def used_rows():
        # rows beyond an environment's distinct targets and states beyond its rail cells are never written
        d = b.dist.view(E, A, -1)
        slot_ok = torch.arange(A, device=d.device)[None, :] < b.n_slots[:, None]
        state_ok = torch.arange(d.shape[2], device=d.device)[None, :] < (b.n_cells * 4)[:, None]
        return torch.where(slot_ok[:, :, None] & state_ok[:, None, :], d, torch.zeros_like(d))
    ref_dist = used_rows()
    stages = args.stages.split(",")
    out = {"workload": args.workload, "E": E, "A": A, "Rp": int(b.Rp), "T1": b.T1, "n_slots_mean": float(b.n_slots.float().mean())}
    if "graph" in stages:
        out["graph_ms"] = timed(lambda: b.build_graph(sync=False))
    if "bfs" in stages:
        for m in ("chains", "bfs"):
            out["dist_ms_" + m] = timed(lambda: b.compute_distances(method=m))
            out["dist_equal_" + m] = bool(torch.equal(used_rows(), ref_dist))
        out["n_anchor_mean_max"] = [float(b.n_anchor.float().mean()), float(b.n_anchor.max()), b.lds_anchor]
        b.compute_distances()
    if "paths" in stages:
        out["paths_ms"] = timed(b.shortest_paths)
    if "bitmaps" in stages:
        out["bitmaps_ms"] = timed(lambda: b.bitmaps(shift=1))
        out["bitmaps_equal"] = bool(torch.equal(b.maps, ref_maps))
        out["bitmaps_GBs"] = E * A * b.plane_stride / out["bitmaps_ms"][0] / 1e6
    if "fused" in stages:
        out["columns_ms"] = timed(lambda: b.columns(shift=1))
        b.obs.zero_()
        t = timed(b.observations_fused)
        out["obs_fused"] = dict(ms=t, GBs_alg=4.0 * b.Rp * b.T1 * E * A / t[0] / 1e6,
                                GBs_written=3.0 * b.plane_stride * E * A / t[0] / 1e6,
                                ok=int(b.obs.view(torch.int64).sum().item()) == ref_obs_sum)
    if "obs" in stages:
        alg = 4.0 * b.Rp * b.T1 * E * A
        b.obs.zero_()
        t = timed(b.observations)
        out["obs_dense"] = dict(ms=t, GBs=alg / t[0] / 1e6, ok=int(b.obs.view(torch.int64).sum().item()) == ref_obs_sum)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
