"""Timings of the DQN kernels on one GPU (development tool): act (B=1), forward and learn at B=512, from float
states and from the uint8 observations a device replay memory holds."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flatland_marl_amd.agent import DQNAgent  # noqa: E402


def timed(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-rails", type=int, default=100)
    ap.add_argument("--depth", type=int, default=150)
    ap.add_argument("--batch", type=int, default=512)
    a = ap.parse_args()
    H3, T1, B = 3 * a.max_rails, a.depth + 1, a.batch
    args = argparse.Namespace(prediction_depth=a.depth, lr=5e-5, buffer_size=10, batch_size=B, update_every=10,
                              gamma=0.99, tau=1e-3)
    torch.manual_seed(0)
    agent = DQNAgent(args, action_space=2, bitmap_height=H3)
    g = torch.Generator().manual_seed(0)
    s = torch.randint(-1, 3, (B, 1, H3, T1), generator=g).float().cuda()
    sn = torch.randint(-1, 3, (B, 1, H3, T1), generator=g).float().cuda()
    act = torch.randint(0, 2, (B,), generator=g).cuda()
    r = torch.randn(B, generator=g).cuda()
    d = (torch.rand(B, generator=g) < 0.1).float().cuda()
    fwd_flop = (2.0 * H3 * T1 * 64 + 4.0 * 64 * H3 * 512) * B
    out = {"H3": H3, "T1": T1, "B": B, "params": agent.qnetwork_local.flat_params().numel()}
    one = s[:1].contiguous()
    t = timed(lambda: agent.qnetwork_local(one), reps=50)
    out["act_b1_ms"] = t
    out["act_b1_weight_GBs"] = out["params"] * 4 / t / 1e6
    t = timed(lambda: agent.qnetwork_local(s))
    out["forward_ms"] = t
    out["forward_TFLOPs"] = fwd_flop / t / 1e9
    ps = (a.max_rails * T1 + 255) // 256 * 256
    obs = torch.randint(0, 3, (B, 3, ps), generator=g, dtype=torch.uint8).cuda()
    t = timed(lambda: agent.qnetwork_local.forward_obs(obs, ps))
    out["forward_u8_ms"] = t
    out["forward_u8_TFLOPs"] = fwd_flop / t / 1e9
    t = timed(lambda: agent.learn((s, act, r, sn, d), 0.99))
    out["learn_ms"] = t
    out["learn_TFLOPs_5x_forward"] = 5 * fwd_flop / t / 1e9
    obs_n = torch.randint(0, 3, (B, 3, ps), generator=g, dtype=torch.uint8).cuda()
    t = timed(lambda: agent.learn_obs((obs, act, r, obs_n, d), 0.99, ps))
    out["learn_u8_ms"] = t
    both = torch.cat([obs, obs_n])                             # the replay memory's layout: next states behind the states
    t = timed(lambda: agent.learn_obs((both[:B], act, r, both[B:], d), 0.99, ps))
    out["learn_u8_adjacent_ms"] = t
    print(json.dumps(out))


if __name__ == "__main__":
    main()
