"""Development tool: replay one golden rollout trace step by step with host-side checks of every index the
apply kernel consumes, printing progress (python tools/debug_rollout.py <trace.npz>)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden_util import load_env  # noqa: E402
from tests.test_gpu_rollout_batched import _rollout  # noqa: E402
from flatland_marl_amd.rollout import hash_policy  # noqa: E402


def main():
    path = sys.argv[1]
    gold = dict(np.load(path))
    name = os.path.basename(path).split("_a")[0]
    A, depth, steps, Rp = int(gold["agents"]), int(gold["depth"]), int(gold["n_steps_run"]), int(gold["max_rails"])
    ro = _rollout(load_env(name), A, depth, Rp, 3)
    b = ro.b
    T1 = b.T1

    def policy(rollout, info):
        net, best = hash_policy(rollout, info)
        torch.cuda.synchronize()
        cols = rollout.cols.cpu().numpy()
        assert cols.min() >= -1 and cols.max() < 2 * Rp, ("cols", cols.min(), cols.max())
        tpc = b.tpc.cpu().numpy()
        assert tpc.min() >= 1 and tpc.max() < T1, ("tpc", tpc.min(), tpc.max())
        br = rollout.branch.cpu().numpy()
        assert br.min() >= 0 and br.max() <= 4
        head, ln = rollout.path_head.cpu().numpy(), rollout.path_len.cpu().numpy()
        assert head.min() >= 0 and ln.max() <= rollout.Lp and ln.min() >= -1, ("path", head.min(), ln.min(), ln.max())
        slot = rollout.alt_slot.cpu().numpy()
        alt = info["alt"]
        if alt is not None:
            n = alt["count"].numel()
            cnt = alt["count"].cpu().numpy()
            assert slot.max() < n and cnt.min() >= 0 and cnt.max() <= rollout.cap_alt, ("alt", slot.max(), n, cnt)
            al = alt["len"].cpu().numpy()
            assert al.min() >= 0 and al.max() <= rollout.Lp, ("alt_len", al.min(), al.max())
            ac = alt["cols"].cpu().numpy()
            used = np.arange(rollout.cap_alt)[None, :] < cnt[:, None]
            assert ac[used].min() >= -1 and ac[used].max() < 2 * Rp, ("alt cols", ac[used].min(), ac[used].max())
            bs = best.cpu().numpy()[slot >= 0]
            assert bs.min() >= 0 and (bs < np.maximum(cnt[slot[slot >= 0]], 1)).all(), ("best", bs)
        else:
            assert slot.max() < 0
        pos = rollout.pos.cpu().numpy()
        assert pos.min() >= -1 and pos.max() < b.H * b.W
        print("step", info["t"], "inputs ok; branch counts", np.bincount(br.ravel(), minlength=5).tolist(), flush=True)
        return net, best

    for t in range(steps):
        ro.step(policy)
        torch.cuda.synchronize()
        for k, got in (("branch", ro.branch), ("net", ro.net), ("crash", ro.crash), ("action", ro.action),
                       ("status", ro.status), ("pos", ro.pos)):
            g = got.cpu().numpy()
            if not np.array_equal(g[0], gold[k][t]):
                print("MISMATCH", k, "step", t, "\n got ", g[0].tolist(), "\n want", gold[k][t].tolist(), flush=True)
                return 1
    print("trace replayed", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
