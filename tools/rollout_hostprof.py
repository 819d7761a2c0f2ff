"""Where the host time of a rollout step goes: cProfile over act-only steps of the 1024 x 10 workload.

    python tools/rollout_hostprof.py [--envs E] [--steps N]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import flatland_marl_amd as fm  # noqa: E402
from flatland_marl_amd.agent import DQNAgent  # noqa: E402
from flatland_marl_amd.rollout import RolloutBatch  # noqa: E402
from flatland_marl_amd.train import BatchedTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=0)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--train", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(0)      # the same weights, hence the same decisions and request counts, on every run
    data, E, A, depth, _ = bench.workload_batch("cfg2g", args.envs)
    b = fm.ObservationBatch(data["grid"], data["start"], data["direction"], data["target"], data["speed"], depth, max_rails=70)
    ro = RolloutBatch(b, data["speed"], data["direction"], max_alt_depth=203,
                      malfunction=dict(rate=2000, min_duration=0, max_duration=0, seed=0)).reset()
    a = types.SimpleNamespace(prediction_depth=depth, lr=5e-5, buffer_size=20000, batch_size=512, update_every=10,
                              gamma=0.99, tau=1e-3)
    tr = BatchedTrainer(ro, DQNAgent(a, action_space=2, bitmap_height=3 * b.Rp), start_eps=0.1, max_learns_per_step=4)
    for _ in range(5):
        tr.step(train=args.train)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(args.steps):
        tr.step(train=args.train)
    torch.cuda.synchronize()
    pr.disable()
    print("ms per step %.3f" % ((time.perf_counter() - t0) / args.steps * 1e3))
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
