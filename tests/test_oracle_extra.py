"""CPU checker against the reference's prediction tables and rail-reordered observations
(tests/golden/extra, produced by oracle/gen_golden_extra.py from the reference's own modules)."""
import glob
import os

import numpy as np
import pytest

from tests.golden_util import GOLDEN, load_env, dense_from_runs

EXTRA = sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(GOLDEN, "extra", "*.npz")))


def load_extra(name):
    return dict(np.load(os.path.join(GOLDEN, "extra", name + ".npz")))


def test_extra_fixtures_present():
    assert len(EXTRA) >= 4


@pytest.mark.parametrize("name", EXTRA)
def test_prediction_cells(name):
    from oracle import oracle as orc
    fx, ex = load_env(name), load_extra(name)
    depth = int(ex["depth"])
    env = orc.OracleEnv(fx, depth)
    for a in range(env.A):
        rows = env.path(a)
        cells = orc.prediction_cells(rows, env.start[a, 0], env.start[a, 1], env.target[a, 0], env.target[a, 1],
                                     env.tpc[a], depth)
        assert np.array_equal(cells, ex["pred"][a][:, 1:3]), (name, a)


@pytest.mark.parametrize("name", EXTRA)
def test_reorder_rails(name):
    from oracle import oracle as orc
    fx, ex = load_env(name), load_extra(name)
    depth = int(ex["depth"])
    Rp = int(ex["max_rails"])
    env = orc.OracleEnv(fx, depth)
    maps, _, errs = env.get_many()
    assert not any(errs)
    seen_hang = False
    for i, a in enumerate(ex["reorder_agents"]):
        if ex["reorder_hang"][i]:
            seen_hang = True
            with pytest.raises(RuntimeError):
                orc.get_obs(maps, a, maps[a], Rp, reorder=True)
            continue
        got = orc.get_obs(maps, a, maps[a], Rp, reorder=True).reshape(3, Rp, depth + 1)
        want = dense_from_runs(ex["reorder_obs_runs_%d" % i], (3, Rp, depth + 1), np.int16)
        assert np.array_equal(got, want), (name, a)
    if name == "Test_3_Level_1":
        assert seen_hang            # the non-terminating case of the reference is covered
