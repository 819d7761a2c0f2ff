import os, sys, glob
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden_util import GOLDEN, load_env
from tests.test_oracle_rollout import shim_env
from oracle.trace_driver import run_trace
from flatland_marl_amd.rail_observations import RailObsForRailEnv
from flatland_marl_amd.predictions import ShortestPathPredictorForRailEnv
from flatland_marl_amd.preprocessing import ObsPreprocessor
path = sorted(glob.glob(os.path.join(GOLDEN, "rollout", "*.npz")))[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
gold = dict(np.load(path))
name = os.path.basename(path).split("_a")[0]
A, depth, steps = int(gold["agents"]), int(gold["depth"]), int(gold["steps"])
ob = RailObsForRailEnv(ShortestPathPredictorForRailEnv(depth))
env = shim_env(load_env(name), ob, A)
import ctypes as C
from oracle import oracle as orc
_orig = ob.update_bitmaps
def checked(a, maps, is_before_switch=False):
    m = np.ascontiguousarray(maps, np.int8).copy()
    st = ob.predictor.device_state()
    status = np.array([int(x.status) for x in env.agents], np.int32)
    tpc = np.array([ob.tpc[i] for i in range(len(env.agents))], np.int32)
    vp = C.c_void_p
    orc.lib().orc_update_bitmaps(m.ctypes.data_as(vp), m.shape[0], m.shape[1], m.shape[2], status.ctypes.data_as(vp), tpc.ctypes.data_as(vp), int(a), int(bool(is_before_switch)))
    before = maps.copy()
    out = _orig(a, maps, is_before_switch)
    if not np.array_equal(out.astype(np.int8), m):
        d = np.nonzero(out.astype(np.int8) != m)
        print("update_bitmaps mismatch agent", a, "before_switch", is_before_switch, "n diff", len(d[0]), "first", [x[:8] for x in d])
        print("before row", before[d[0][0], d[1][0], :12], "got", out[d[0][0], d[1][0], :12], "want", m[d[0][0], d[1][0], :12])
        sys.exit(0)
    return out
ob.update_bitmaps = checked
rec, snaps = run_trace(env, ob, ObsPreprocessor(int(gold["max_rails"]), False), steps, hash_obs=False)
for s in range(int(gold["n_steps_run"])):
    for k in ("branch", "net", "crash", "action", "best", "n_alt", "status", "pos"):
        if not np.array_equal(rec[k][s], gold[k][s]):
            bad = np.nonzero(rec[k][s] != gold[k][s])[0]
            print("step", s, "key", k, "agents", bad[:10], "got", rec[k][s][bad[:10]], "want", gold[k][s][bad[:10]])
            for kk in ("branch", "net", "crash", "action", "best", "n_alt"):
                print("   ", kk, rec[kk][s][bad[:5]], gold[kk][s][bad[:5]])
            sys.exit(0)
    if rec["maps_sha"][s] != gold["maps_sha"][s]:
        print("step", s, "maps differ")
        sys.exit(0)
print("all equal")
