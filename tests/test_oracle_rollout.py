"""The CPU checker replays the golden rollout traces recorded with the reference's builder."""
import glob
import os
import sys

import numpy as np
import pytest

from tests.golden_util import GOLDEN, load_env, dense_from_runs

SHIM = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "flatland_shim")
TRACES = sorted(glob.glob(os.path.join(GOLDEN, "rollout", "*.npz")))


def malfunction_of(gold):
    """The malfunction process a trace was recorded with (None for most)."""
    if "malfunction" not in gold:
        return None
    rate, lo, hi, seed = (int(x) for x in gold["malfunction"])
    return dict(rate=rate, min_duration=lo, max_duration=hi, seed=seed)


def shim_env(fx, builder, A, malfunction=None):
    if SHIM not in sys.path:
        sys.path.insert(0, os.path.abspath(SHIM))
    from flatland.envs.rail_env import RailEnv
    return RailEnv.from_fixture(fx, builder, num_agents=A, malfunction=malfunction)


def compare_trace(rec, snaps, gold):
    n = int(gold["n_steps_run"])
    assert int(rec["n_steps_run"]) == n
    for k in ("branch", "net", "crash", "action", "best", "n_alt", "status", "pos", "malf"):
        if k in gold:
            assert np.array_equal(rec[k], gold[k]), k
    assert list(rec["maps_sha"]) == list(gold["maps_sha"])
    assert str(rec["obs_sha"]) == str(gold["obs_sha"])
    for s in gold["snap_steps"]:
        want = dense_from_runs(gold["snap_%d" % s], snaps[int(s)].shape)
        assert np.array_equal(snaps[int(s)], want)


@pytest.mark.parametrize("path", TRACES, ids=[os.path.basename(p)[:-4] for p in TRACES])
def test_oracle_replays_trace(path):
    from oracle.oracle_builder import OracleRailObs, OraclePreprocessor
    from oracle.trace_driver import run_trace
    gold = dict(np.load(path))
    name = os.path.basename(path).split("_a")[0]
    A, depth, steps = int(gold["agents"]), int(gold["depth"]), int(gold["steps"])
    ob = OracleRailObs(depth)
    env = shim_env(load_env(name), ob, A, malfunction_of(gold))
    rec, snaps = run_trace(env, ob, OraclePreprocessor(int(gold["max_rails"])), steps)
    compare_trace(rec, snaps, gold)
