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
    # grids from the package's own generator (tests/golden/make_generated_fixtures.py)
    ("Gen_25x25_0", 10, 150, 150),
    ("Gen_64x64_3", 40, 150, 199),
]
# the same driver with the stand-in's malfunction process switched on: (fixture, agents, depth, steps,
# rate, min_duration, max_duration, seed).  The reference's loop never looks at malfunctions, so its
# occupancy maps run ahead of broken agents until the next switch or the end of a path re-plans them.
MALFUNCTION_CASES = [
    ("Test_0_Level_0", 50, 150, 150, 25, 0, 0, 11),
    ("Test_2_Level_0", 40, 150, 150, 40, 1, 4, 12),
    ("Test_3_Level_0", 30, 150, 199, 15, 0, 6, 13),
]


def main():
    out_dir = os.path.join(ROOT, "tests", "golden", "rollout")
    os.makedirs(out_dir, exist_ok=True)
    only_malfunction = "--malfunction-only" in sys.argv
    cases = [] if only_malfunction else [c + (None,) for c in CASES]
    cases += [c[:4] + (dict(rate=c[4], min_duration=c[5], max_duration=c[6], seed=c[7]),) for c in MALFUNCTION_CASES]
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    for name, A, depth, steps, malf in cases:
        if only and name not in only:
            continue
        t0 = time.time()
        fx = np.load(os.path.join(ROOT, "tests", "golden", "envs", name + ".npz"))
        ob = RailObsForRailEnv(ShortestPathPredictorForRailEnv(depth))
        env = RailEnv.from_fixture(fx, ob, num_agents=A, malfunction=malf)
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
        if malf:
            rec["malfunction"] = np.array([malf["rate"], malf["min_duration"], malf["max_duration"], malf["seed"]],
                                          np.int64)
        np.savez_compressed(os.path.join(out_dir, "%s_a%d_d%d%s.npz" % (name, A, depth, "_m" if malf else "")), **rec)
        br = rec["branch"]
        print(name, A, "steps run", int(rec["n_steps_run"]), "branches", np.bincount(br.ravel(), minlength=5),
              "crashes", int(rec["crash"].sum()), "done", int((rec["status"][-1] >= 2).sum()),
              "alts>1", int((rec["n_alt"] > 1).sum()), "broken agent-steps", int((rec["malf"] > 0).sum()),
              "%.1fs" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
