"""The CPU checker against vectors produced by the reference's own modules."""
import hashlib

import numpy as np
import pytest

from oracle import oracle as orc
from tests.golden_util import (golden_cases, load_env, load_golden, dense_from_runs, sha, path_rows_of)

CASES = golden_cases()


@pytest.mark.parametrize("name,depth", CASES)
def test_graph(name, depth):
    if depth != 150:
        pytest.skip("graph does not depend on depth")
    fx, g = load_env(name), load_golden(name, depth)
    gr = orc.Graph(fx["grid"])
    assert gr.N == g["node_cells"].shape[0] and gr.R == g["info"].shape[0]
    assert np.array_equal(np.stack([gr.node_cell // gr.W, gr.node_cell % gr.W], 1), g["node_cells"])
    assert np.array_equal(gr.edge_info, g["info"])
    assert np.array_equal(gr.edge_cell_off, g["edge_cell_off"])
    ec = gr.edge_cells
    mine = np.stack([(ec >> 2) // gr.W, (ec >> 2) % gr.W, ec & 3], 1)
    assert np.array_equal(mine, g["edge_cells"])


@pytest.mark.parametrize("name,depth", CASES)
def test_distance_and_paths(name, depth):
    fx, g = load_env(name), load_golden(name, depth)
    env = orc.OracleEnv(fx, depth)
    if depth == 150:
        dm = np.stack([env.dist(a) for a in range(env.A)])
        assert sha(dm) == str(g["dist_sha"])
    for a in range(env.A):
        rows = env.path(a)
        if g["path_none"][a]:
            assert rows is None
        else:
            assert np.array_equal(rows, path_rows_of(g, a)), (name, a)


@pytest.mark.parametrize("name,depth", CASES)
def test_get_many_and_obs(name, depth):
    fx, g = load_env(name), load_golden(name, depth)
    env = orc.OracleEnv(fx, depth)
    maps, paths, errs = env.get_many()
    err = str(g["error"])
    if err:
        first = [e for e in errs if e][0]
        assert first == err
        return
    assert not any(errs)
    want = dense_from_runs(g["maps_runs"], tuple(g["maps_shape"]))
    assert np.array_equal(maps, want)
    assert sha(maps) == str(g["maps_sha"])
    Rp = int(g["obs_rp"])
    h = hashlib.sha256()
    samples = list(g["obs_sample_agents"])
    for a in range(env.A):
        o = orc.get_obs(maps, a, maps[a], Rp)
        h.update(o.tobytes())
        if a in samples:
            w = dense_from_runs(g["obs_sample_runs_%d" % samples.index(a)], (3, Rp, depth + 1), np.int16)
            assert np.array_equal(o.reshape(3, Rp, depth + 1), w)
    assert h.hexdigest() == str(g["obs_sha"])


@pytest.mark.parametrize("name,depth", [c for c in CASES if c[1] == 150])
def test_altmaps(name, depth):
    fx, g = load_env(name), load_golden(name, depth)
    if str(g["error"]):
        pytest.skip("reference raises on this map")
    env = orc.OracleEnv(fx, depth)
    n_alt_total = int(np.maximum(g["alt_count"], 0).sum())
    want_maps = dense_from_runs(g["alt_maps_runs"], (n_alt_total, env.g.R, depth + 1))
    flat = 0
    for a, k, cnt in zip(g["alt_agent"], g["alt_k"], g["alt_count"]):
        el = path_rows_of(g, a)[k]
        bms, paths, err = env.altmaps(a, el[0], el[1], el[2])
        if cnt < 0:
            assert err
            continue
        assert not err and len(bms) == cnt, (name, a, err)
        for i in range(cnt):
            w = g["alt_path_rows"][g["alt_path_off"][flat]:g["alt_path_off"][flat + 1]].astype(np.int32)
            assert np.array_equal(paths[i], w), (name, a, i)
            assert np.array_equal(bms[i], want_maps[flat]), (name, a, i)
            flat += 1
