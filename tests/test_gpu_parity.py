"""HIP path vs the CPU checker and the committed goldens, through the C ABI (needs an MI355X)."""
import numpy as np
import pytest

from tests.golden_util import env_names, load_env, load_golden, dense_from_runs, sha, path_rows_of

pytestmark = pytest.mark.gpu

ERR = {"IndexError": 2, "KeyError": 3, "AssertionError": 4}


def _batch(fx, depth, **kw):
    import flatland_marl_amd as fm
    return fm.ObservationBatch(fx["grid"], fx["initial_position"], fx["direction"], fx["target"], fx["speed"], depth, **kw)


@pytest.mark.parametrize("name", env_names())
def test_graph_tables(name):
    from oracle import oracle as orc
    fx = load_env(name)
    b = _batch(fx, 150).build_graph()
    g = orc.Graph(fx["grid"])
    assert int(b.num_nodes[0]) == g.N and int(b.R[0]) == g.R
    assert np.array_equal(b.node_id[0].cpu().numpy(), g.node_id)
    assert np.array_equal(b.node_cell[0, :g.N].cpu().numpy(), g.node_cell)
    assert np.array_equal(b.edge_info[0, :g.R].cpu().numpy(), g.edge_info)
    assert np.array_equal(b.edge_cell_off[0, :g.R + 1].cpu().numpy(), g.edge_cell_off)
    assert np.array_equal(b.edge_cells[0, :g.edge_cells.shape[0]].cpu().numpy(), g.edge_cells)
    assert np.array_equal(b.cell_rail[0].cpu().numpy(), g.cell_rail)


# The two routes from a reset to the planes and observations.  "dense": distance maps -> shortest paths -> planes
# -> observation kernel over dense planes (reset() + observations()).  "fused": the route the throughput figure is
# measured on, reset_fused(): paths and columns straight from the chain index (flm_paths_columns) wherever the
# tables fit the LDS, then the observations straight from the column form (flm_observations_columns).  Both meet
# the reference's own outputs here.
ROUTES = ["dense", "fused"]


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("name", env_names())
def test_distances_paths_maps_obs(name, route):
    import torch
    from oracle import oracle as orc
    depth = 150
    fx, gold = load_env(name), load_golden(name, depth)
    b = _batch(fx, depth)
    env = orc.OracleEnv(fx, depth)
    if route == "fused":
        obs = b.reset_fused()
        b.check_graph()
        # plane 0 of every observation is the dense occupancy plane
        b.maps = obs[:, :, 0].contiguous().view(torch.int8)
    else:
        b.reset()
        # distances: compact uint16 -> dense int32, against the checker (and the golden hash)
        cell_idx = b.cell_idx[0].cpu().numpy()
        dist = b.dist[0].cpu().numpy().view(np.uint16)
        tslot = b.tslot[0].cpu().numpy()
        rail = cell_idx >= 0
        dense = np.full((env.A, env.H * env.W, 4), -1, np.int32)
        for a in range(env.A):
            d = dist[tslot[a]].reshape(-1, 4)[cell_idx[rail]].astype(np.int32)
            d[d == 0xFFFF] = -1
            dense[a, rail] = d
        dense = dense.reshape(env.A, env.H, env.W, 4)
        for a in range(env.A):
            assert np.array_equal(dense[a], env.dist(a)), (name, a)
        assert sha(dense) == str(gold["dist_sha"])
    # shortest paths against the golden (reference output)
    paths = b.paths_numpy(0)
    for a in range(env.A):
        if gold["path_none"][a]:
            assert paths[a] is None
        else:
            assert np.array_equal(paths[a], path_rows_of(gold, a)), (name, a)
    # occupancy planes
    err = b.agent_err[0].cpu().numpy()
    gerr = str(gold["error"])
    omaps, _, oerrs = env.get_many()
    assert [ERR.get(x, 0) for x in oerrs] == list(err)
    if gerr:
        first = [x for x in err if x][0]
        assert first == ERR[gerr]
        return
    want = dense_from_runs(gold["maps_runs"], tuple(gold["maps_shape"]))
    got = b.maps_numpy(0)
    assert np.array_equal(got, want)
    assert sha(got) == str(gold["maps_sha"])
    # padded rows and the tail of each plane stay zero
    full = b.maps[0].cpu().numpy()
    assert not full[:, int(b.R[0]) * b.T1:].any()


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("name", [n for n in env_names() if not str(load_golden(n)["error"])])
def test_observations_golden(name, route):
    import hashlib
    depth = 150
    fx, gold = load_env(name), load_golden(name, depth)
    Rp = int(gold["obs_rp"])
    b = _batch(fx, depth, max_rails=Rp)
    if route == "fused":
        b.reset_fused()
    else:
        b.reset()
        b.observations()
    obs = b.obs_numpy(0)
    assert obs.shape == (b.A, 3, Rp, depth + 1)
    h = hashlib.sha256()
    for a in range(b.A):
        h.update(obs[a].tobytes())
    assert h.hexdigest() == str(gold["obs_sha"])
    for i, a in enumerate(gold["obs_sample_agents"]):
        w = dense_from_runs(gold["obs_sample_runs_%d" % i], (3, Rp, depth + 1), np.int16)
        assert np.array_equal(obs[a], w)


def test_depth300_large_map():
    depth = 300
    fx, gold = load_env("Test_9_Level_0"), load_golden("Test_9_Level_0", depth)
    b = _batch(fx, depth)
    b.reset()
    want = dense_from_runs(gold["maps_runs"], tuple(gold["maps_shape"]))
    assert np.array_equal(b.maps_numpy(0), want)
    import hashlib
    Rp = int(gold["obs_rp"])
    b2 = _batch(fx, depth, max_rails=Rp)
    b2.reset()
    b2.observations()
    obs = b2.obs_numpy(0)
    h = hashlib.sha256()
    for a in range(b2.A):
        h.update(obs[a].tobytes())
    assert h.hexdigest() == str(gold["obs_sha"])


def test_batched_envs_match_single():
    """E > 1: every environment of a batch equals the same environment run alone (agents permuted per env)."""
    from oracle import oracle as orc
    depth = 150
    fx = load_env("Test_2_Level_0")
    E, A = 5, 80
    rng = np.random.default_rng(0)
    perms = [rng.permutation(A) for _ in range(E)]
    import flatland_marl_amd as fm
    grid = np.stack([fx["grid"]] * E)
    b = fm.ObservationBatch(grid, np.stack([fx["initial_position"][p] for p in perms]),
                            np.stack([fx["direction"][p] for p in perms]),
                            np.stack([fx["target"][p] for p in perms]),
                            np.stack([fx["speed"][p] for p in perms]), depth, max_rails=100)
    b.reset()
    b.observations()
    env = orc.OracleEnv(fx, depth)
    maps, _, errs = env.get_many()
    for e in range(E):
        got = b.maps_numpy(e)
        assert np.array_equal(got, maps[perms[e]])
        obs = b.obs_numpy(e)
        for a in (0, 17, 79):
            want = orc.get_obs(maps[perms[e]], a, maps[perms[e]][a], 100).reshape(3, 100, depth + 1)
            assert np.array_equal(obs[a], want)


@pytest.mark.parametrize("name,depth", [("Test_0_Level_0", 150), ("Test_3_Level_0", 150), ("Test_6_Level_0", 150),
                                        ("Test_8_Level_1", 150), ("Test_9_Level_1", 300), ("Test_1_Level_1", 40)])
def test_fused_observations_equal_unfused(name, depth):
    """Column form + fused kernel write exactly the bytes of planes + observation kernel (errors included)."""
    import torch
    fx = load_env(name)
    b = _batch(fx, depth)
    b.reset()
    want = b.observations().clone()
    err = b.agent_err.clone()
    want_maps = b.maps.clone()
    b.obs = None
    got = b.reset_fused()
    assert torch.equal(b.agent_err, err)
    assert torch.equal(got, want)
    # plane 0 of the observation is the dense occupancy plane
    assert torch.equal(got[:, :, 0].contiguous().view(torch.int8), want_maps)


def test_long_horizon_uses_sequential_formulation():
    """depth 600 (> 512 time steps) takes the one-workgroup-per-path kernel; same planes as the checker."""
    from oracle import oracle as orc
    depth = 600
    fx = load_env("Test_1_Level_0")
    b = _batch(fx, depth)
    b.reset()
    env = orc.OracleEnv(fx, depth)
    maps, _, errs = env.get_many()
    assert [ERR.get(x, 0) for x in errs] == list(b.agent_err[0].cpu().numpy())
    assert np.array_equal(b.maps_numpy(0), maps)


def test_long_horizon_observations_fall_back_to_dense_planes():
    """reset_fused at depth 600: the column kernels stop at 512 time steps, the observations come from the dense
    planes instead and are the checker's."""
    from oracle import oracle as orc
    depth = 600
    fx = load_env("Test_1_Level_0")
    b = _batch(fx, depth)
    obs = b.reset_fused()
    env = orc.OracleEnv(fx, depth)
    maps, _, errs = env.get_many()
    assert [ERR.get(x, 0) for x in errs] == list(b.agent_err[0].cpu().numpy())
    R, T1, A = env.g.R, depth + 1, b.A
    got = obs[0, :, :, :b.Rp * T1].cpu().numpy().reshape(A, 3, b.Rp, T1)
    assert np.array_equal(got[:, 0, :R].view(np.int8), maps)
    tot_p, tot_n = (maps > 0).sum(axis=0), (maps < 0).sum(axis=0)
    for a in range(A):
        assert np.array_equal(got[a, 1, :R], tot_p - (maps[a] > 0))
        assert np.array_equal(got[a, 2, :R], tot_n - (maps[a] < 0))


@pytest.mark.parametrize("name", ["Test_1_Level_0", "Test_5_Level_1", "Test_8_Level_0"])
def test_breadth_first_fallback_equals_chain_compression(name):
    """The two distance methods give the same value for every valid state of every used row."""
    import torch
    fx = load_env(name)
    b = _batch(fx, 150)
    b.reset()
    d0 = b.dist.clone()
    b.dist.fill_(12345)
    b.compute_distances(method="bfs")
    torch.cuda.synchronize()
    S = d0.shape[2]
    valid = (torch.arange(S, device=d0.device)[None, None, :] < (b.n_cells * 4)[:, None, None]) & \
            (torch.arange(b.A, device=d0.device)[None, :, None] < b.n_slots[:, None, None])
    assert bool((d0[valid] == b.dist[valid]).all())
