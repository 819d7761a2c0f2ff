"""Rollout throughput on one GPU (development tool): environment steps per second of the batched driver,
with the hash policy (bookkeeping + environment only) and with the DQN policy (observations + network),
optionally with learning.

    python tools/rollout_bench.py [--workload cfg2] [--envs E] [--steps N] [--max-rails R] [--train]
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import flatland_marl_amd as fm  # noqa: E402
from flatland_marl_amd.agent import DQNAgent  # noqa: E402
from flatland_marl_amd.rollout import RolloutBatch, hash_policy  # noqa: E402
from flatland_marl_amd.train import BatchedTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--envs", type=int, default=0)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--max-rails", type=int, default=0)
    ap.add_argument("--train", action="store_true")
    ap.add_argument("--eps", type=float, default=0.1)
    ap.add_argument("--alt-depth", type=int, default=500)
    ap.add_argument("--malfunction-rate", type=int, default=2000, help="0 switches the malfunction process off")
    ap.add_argument("--min-duration", type=int, default=0)
    ap.add_argument("--max-duration", type=int, default=0)
    args = ap.parse_args()
    data, E, A, depth, desc = bench.workload_batch(args.workload, args.envs)
    b = fm.ObservationBatch(data["grid"], data["start"], data["direction"], data["target"], data["speed"], depth,
                            max_rails=args.max_rails or None)
    malf = dict(rate=args.malfunction_rate, min_duration=args.min_duration, max_duration=args.max_duration, seed=0)
    ro = RolloutBatch(b, data["speed"], data["direction"], max_alt_depth=args.alt_depth, malfunction=malf).reset()
    torch.cuda.synchronize()
    out = dict(workload=args.workload, E=E, A=A, Rp=int(b.Rp), T1=b.T1)

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    # 1. bookkeeping + environment only
    for _ in range(3):
        ro.step(hash_policy)
    dt = timed(lambda: ro.step(hash_policy), args.steps)
    out["hash_policy"] = dict(ms_per_step=dt * 1e3, env_steps_per_s=E / dt, agent_steps_per_s=E * A / dt)
    st = ro.status.cpu().numpy()
    out["status_after"] = np.bincount(st.ravel(), minlength=4).tolist()
    # kernel-level: apply and environment step alone
    net = torch.ones((E, A), dtype=torch.int32, device=ro.device)
    best = torch.zeros((E, A), dtype=torch.int32, device=ro.device)
    ro.classify()
    ro.alt_slot.fill_(-1)
    out["classify_ms"] = timed(ro.classify, 20) * 1e3
    out["apply_ms"] = timed(lambda: ro.apply(net, best, None), 20) * 1e3
    out["env_step_ms"] = timed(ro.env_step, 20) * 1e3

    # 2. the network in the loop
    ro.reset()
    a = types.SimpleNamespace(prediction_depth=depth, lr=5e-5, buffer_size=20000, batch_size=512, update_every=10,
                              gamma=0.99, tau=1e-3)
    agent = DQNAgent(a, action_space=2, bitmap_height=3 * b.Rp)
    tr = BatchedTrainer(ro, agent, start_eps=args.eps, max_learns_per_step=4)
    for _ in range(3):
        tr.step(train=args.train)
    n0, l0 = tr.n_transitions, tr.n_learn
    dt = timed(lambda: tr.step(train=args.train), args.steps)
    out["dqn_policy"] = dict(ms_per_step=dt * 1e3, env_steps_per_s=E / dt, train=bool(args.train),
                             transitions_per_step=(tr.n_transitions - n0) / args.steps,
                             learns_per_step=(tr.n_learn - l0) / args.steps)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
