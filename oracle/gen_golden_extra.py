"""Extra golden vectors from the reference's own modules (build container only): prediction tables
(src/predictions.py:85-135) and observations with rail reordering (src/preprocessing.py:32-74).

TEST INFRASTRUCTURE ONLY.   python oracle/gen_golden_extra.py
"""
import os
import signal
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "flatland_shim"))
sys.path.insert(0, "/root/reference")

from flatland.envs.rail_env import RailEnv  # noqa: E402
from src.rail_observations import RailObsForRailEnv  # noqa: E402
from src.predictions import ShortestPathPredictorForRailEnv  # noqa: E402
from src.preprocessing import ObsPreprocessor  # noqa: E402
from oracle.gen_golden import runs_of  # noqa: E402

CASES = [("Test_0_Level_0", 150), ("Test_2_Level_0", 150), ("Test_3_Level_1", 60), ("Test_5_Level_0", 150)]


class _Hang(Exception):
    pass


def _alarm(signum, frame):
    raise _Hang()


def main():
    out_dir = os.path.join(ROOT, "tests", "golden", "extra")
    os.makedirs(out_dir, exist_ok=True)
    for name, depth in CASES:
        fx = np.load(os.path.join(ROOT, "tests", "golden", "envs", name + ".npz"))
        ob = RailObsForRailEnv(ShortestPathPredictorForRailEnv(depth))
        env = RailEnv.from_fixture(fx, ob)
        maps, _ = env.reset()
        A = len(env.agents)
        pred = ob.predictor.get()
        out = {"depth": np.array(depth, np.int32)}
        out["pred"] = np.stack([pred[a] for a in range(A)]).astype(np.int16)       # [A, depth+1, 5]
        Rp = max(100, ob.num_rails)
        pp = ObsPreprocessor(Rp, True)
        agents = list(range(0, A, max(1, A // 12)))[:12]
        out["reorder_agents"] = np.array(agents, np.int32)
        out["max_rails"] = np.array(Rp, np.int32)
        obs_runs, bm_after = [], []
        hang = []
        for a in agents:
            bm = maps[a].copy()
            # src/preprocessing.py:49-52 never returns when the current rail stays occupied up to the
            # last time step (np.argmax of an all-False slice is 0): bound it and record the fact.
            signal.signal(signal.SIGALRM, _alarm)
            signal.alarm(3)
            try:
                o = pp.get_obs(a, bm, maps)
                signal.alarm(0)
                hang.append(0)
            except _Hang:
                hang.append(1)
                o = np.zeros((3 * Rp, depth + 1))
                bm = maps[a].copy()
            assert np.array_equal(o, o.astype(np.int16))
            obs_runs.append(runs_of(o.astype(np.int16).reshape(3, Rp, depth + 1)))
            bm_after.append(bm)                       # the reference reorders the caller's array in place
        out["reorder_hang"] = np.array(hang, np.int32)
        for i in range(len(agents)):
            out["reorder_obs_runs_%d" % i] = obs_runs[i]
        out["reorder_bitmap_after_runs"] = runs_of(np.stack(bm_after))
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **out)
        print(name, "pred", out["pred"].shape, "reorder agents", len(agents), "hang", hang)


if __name__ == "__main__":
    main()
