"""Record golden rollout traces with the reference's own builder classes (build container only).

TEST INFRASTRUCTURE ONLY.  See oracle/trace_driver.py for what is recorded.

    python oracle/gen_golden_rollout.py
"""
import os
import sys
import time

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
from oracle.trace_driver import run_trace  # noqa: E402
from oracle.gen_golden import runs_of  # noqa: E402

# (fixture, agents, depth, steps)
CASES = [
    ("Test_0_Level_0", 50, 150, 120),
    ("Test_0_Level_1", 30, 150, 120),
    ("Test_1_Level_0", 40, 150, 150),
    ("Test_2_Level_0", 40, 150, 150),
    ("Test_2_Level_1", 80, 150, 100),
    ("Test_3_Level_0", 30, 150, 150),
    ("Test_5_Level_0", 25, 100, 199),
]


def main():
    out_dir = os.path.join(ROOT, "tests", "golden", "rollout")
    os.makedirs(out_dir, exist_ok=True)
    for name, A, depth, steps in CASES:
        t0 = time.time()
        fx = np.load(os.path.join(ROOT, "tests", "golden", "envs", name + ".npz"))
        ob = RailObsForRailEnv(ShortestPathPredictorForRailEnv(depth))
        env = RailEnv.from_fixture(fx, ob, num_agents=A)
        ob.reset()
        Rp = max(100, ob.num_rails)
        pp = ObsPreprocessor(Rp, False)
        rec, snaps = run_trace(env, ob, pp, steps)
        rec["agents"] = np.array(A, np.int32)
        rec["depth"] = np.array(depth, np.int32)
        rec["steps"] = np.array(steps, np.int32)
        rec["max_rails"] = np.array(Rp, np.int32)
        rec["snap_steps"] = np.array(sorted(snaps), np.int32)
        for s in sorted(snaps):
            rec["snap_%d" % s] = runs_of(snaps[s])
        np.savez_compressed(os.path.join(out_dir, "%s_a%d_d%d.npz" % (name, A, depth)), **rec)
        br = rec["branch"]
        print(name, A, "steps run", int(rec["n_steps_run"]), "branches", np.bincount(br.ravel(), minlength=5),
              "crashes", int(rec["crash"].sum()), "done", int((rec["status"][-1] >= 2).sum()),
              "alts>1", int((rec["n_alt"] > 1).sum()), "%.1fs" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
