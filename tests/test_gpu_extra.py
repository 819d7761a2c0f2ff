"""Prediction tables and rail reordering of the HIP path against the reference's outputs (needs an MI355X)."""
import numpy as np
import pytest

from tests.golden_util import load_env, dense_from_runs
from tests.test_oracle_extra import EXTRA, load_extra
from tests.test_oracle_rollout import shim_env

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", EXTRA)
def test_prediction_tables(name):
    import flatland_marl_amd as fm
    fx, ex = load_env(name), load_extra(name)
    depth = int(ex["depth"])
    ob = fm.RailObsForRailEnv(fm.ShortestPathPredictorForRailEnv(depth))
    env = shim_env(fx, ob, None)
    env.reset()
    tables = ob.predictor.get()
    for a in range(len(env.agents)):
        assert np.array_equal(np.asarray(tables[a]), ex["pred"][a].astype(int)), (name, a)


@pytest.mark.parametrize("name", EXTRA)
def test_reorder_rails_dropin(name):
    import flatland_marl_amd as fm
    fx, ex = load_env(name), load_extra(name)
    depth, Rp = int(ex["depth"]), int(ex["max_rails"])
    b = fm.ObservationBatch(fx["grid"], fx["initial_position"], fx["direction"], fx["target"], fx["speed"], depth)
    b.reset()
    maps = b.maps_numpy(0).astype(np.int64)
    after = dense_from_runs(ex["reorder_bitmap_after_runs"], (len(ex["reorder_agents"]),) + maps.shape[1:], np.int64)
    pp = fm.ObsPreprocessor(Rp, True)
    for i, a in enumerate(ex["reorder_agents"]):
        bm = maps[a].copy()
        if ex["reorder_hang"][i]:
            with pytest.raises(RuntimeError):
                pp.get_obs(int(a), bm, maps)
            continue
        got = pp.get_obs(int(a), bm, maps)
        assert got.dtype == np.float64 and got.shape == (3 * Rp, depth + 1)
        want = dense_from_runs(ex["reorder_obs_runs_%d" % i], (3, Rp, depth + 1), np.int16)
        assert np.array_equal(got.reshape(3, Rp, depth + 1), want), (name, a)
        assert np.array_equal(bm, after[i]), (name, a)       # the caller's bitmap is reordered in place


def test_reorder_rails_batched_matches_checker():
    import torch
    import flatland_marl_amd as fm
    from oracle import oracle as orc
    name = "Test_5_Level_0"
    fx = load_env(name)
    depth = 150
    b = fm.ObservationBatch(fx["grid"], fx["initial_position"], fx["direction"], fx["target"], fx["speed"], depth)
    b.reset()
    obs = b.observations()
    out, err = b.reorder_rails(obs)
    torch.cuda.synchronize()
    err = err.cpu().numpy()
    maps = b.maps_numpy(0)
    got = out[0, :, :, :b.Rp * b.T1].cpu().numpy().reshape(b.A, 3, b.Rp, b.T1)
    same = obs[0, :, :, :b.Rp * b.T1].cpu().numpy().reshape(b.A, 3, b.Rp, b.T1)
    n_bad = 0
    for a in range(b.A):
        try:
            want = orc.get_obs(maps, a, maps[a], b.Rp, reorder=True).reshape(3, b.Rp, b.T1)
        except RuntimeError:
            assert err[a] == 6
            assert np.array_equal(got[a], same[a])
            n_bad += 1
            continue
        assert err[a] == 0
        g = got[a].astype(np.int16)
        g[0] = got[a, 0].view(np.int8)
        assert np.array_equal(g, want), a
    assert 0 < n_bad < b.A
    assert not out[0, :, :, b.Rp * b.T1:].any()
