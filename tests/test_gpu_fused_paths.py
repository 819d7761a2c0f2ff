"""flm_paths_columns (paths + occupancy columns straight from the chain index, no dense distance map) against
flm_distance_chains -> flm_shortest_paths -> flm_columns, which the other GPU tests pin to the reference's outputs:
path lengths, every path element, every column and every error code must be identical.  Needs an MI355X."""
import numpy as np
import pytest

from tests.golden_util import env_names, load_env

pytestmark = pytest.mark.gpu


def _batch(fx, depth, A=None, **kw):
    import flatland_marl_amd as fm
    sl = slice(None) if A is None else slice(0, A)
    return fm.ObservationBatch(fx["grid"], fx["initial_position"][sl], fx["direction"][sl], fx["target"][sl],
                               fx["speed"][sl], depth, **kw)


def _compare(b_ref, b_fus, tag):
    import torch
    torch.cuda.synchronize()
    b_fus.check_graph()
    ln = b_ref.path_len.cpu().numpy()
    assert np.array_equal(ln, b_fus.path_len.cpu().numpy()), tag
    pc, pm = b_ref.path_cell.cpu().numpy(), b_ref.path_meta.cpu().numpy()
    fc, fm_ = b_fus.path_cell.cpu().numpy(), b_fus.path_meta.cpu().numpy()
    k = np.arange(pc.shape[2])[None, None, :] < ln[:, :, None]
    assert np.array_equal(pc[k], fc[k]), tag
    assert np.array_equal(pm[k], fm_[k]), tag
    assert np.array_equal(b_ref.agent_err.cpu().numpy(), b_fus.agent_err.cpu().numpy()), tag
    assert np.array_equal(b_ref.cols.cpu().numpy(), b_fus.cols.cpu().numpy()), tag
    return ln


def _cases():
    out = []
    for n in env_names():
        out.append((n, 150, None))
    out += [("Test_9_Level_0", 300, 60), ("Test_1_Level_1", 40, None), ("Test_4_Level_0", 7, None),
            ("Test_8_Level_0", 150, 50), ("Test_8_Level_1", 150, 50), ("Test_9_Level_1", 150, 50)]
    return out


@pytest.mark.parametrize("name,depth,A", _cases())
def test_fused_equals_dense_path(name, depth, A):
    fx = load_env(name)
    b = _batch(fx, depth, A)
    b.build_graph()
    b.compute_distances()
    b.shortest_paths()
    b.columns(shift=1)
    f = _batch(fx, depth, A)
    f.build_graph()
    f.chain_index()
    if not f.fused_fits():
        pytest.skip("tables of %s (A=%d, depth %d) do not fit the LDS: the dense path is the one that runs" % (name, f.A, depth))
    f.paths_columns_fused(shift=1)
    ln = _compare(b, f, name)
    assert (ln != 0).any()


@pytest.mark.parametrize("name", ["Test_2_Level_0", "Test_5_Level_0", "Test_7_Level_1"])
def test_fused_from_positions_along_the_way(name):
    """Agents somewhere on the grid (rollout state): start cells and headings taken from their own paths, mixed
    statuses (one in eight DONE_REMOVED), no shift, a shorter walk than the horizon."""
    fx = load_env(name)
    depth = 150
    b = _batch(fx, depth, 40)
    b.build_graph()
    b.compute_distances()
    b.shortest_paths()
    pc = b.path_cell[0].cpu().numpy()
    pm = b.path_meta[0].cpu().numpy().view(np.uint32)
    ln = b.path_len[0].cpu().numpy()
    rng = np.random.default_rng(5)
    A = b.A
    pos = b.pos_cell[0].cpu().numpy().copy()
    hd = b.direction[0].cpu().numpy().copy()
    st = np.ones(A, np.int32)
    for a in range(A):
        if ln[a] > 2:
            k = int(rng.integers(0, ln[a] - 1))
            pos[a], hd[a] = pc[a, k], pm[a, k] & 0xFF
        if a % 8 == 7:
            st[a] = 3
        elif a % 8 == 0:
            st[a] = 0
    f = _batch(fx, depth, 40)
    for x in (b, f):
        x.set_agent_state(status=st, pos_cell=pos, direction=hd)
    b.shortest_paths(max_depth=90)
    b.columns(shift=0)
    f.build_graph()
    f.chain_index()
    assert f.fused_fits(max_depth=90)
    f.paths_columns_fused(shift=0, max_depth=90)
    _compare(b, f, name)


def test_fused_batch_of_generated_grids():
    """Many environments per launch, grids of the package's generator (dead ends excluded by construction, several
    agents sharing targets), agents per environment not a multiple of anything."""
    import flatland_marl_amd as fm
    from flatland_marl_amd.railgen import make_generated_batch
    data = make_generated_batch(96, 13, 30, 30, seed=3, n_maps=6, depth=100, spacing=4)
    data["target"][:, 5] = data["target"][:, 2]            # shared targets: fewer distance rows than agents
    mk = lambda: fm.ObservationBatch(data["grid"], data["start"], data["direction"], data["target"], data["speed"], 100)
    b, f = mk(), mk()
    b.build_graph()
    b.compute_distances()
    b.shortest_paths()
    b.columns(shift=1)
    f.build_graph()
    f.chain_index()
    assert f.fused_fits()
    f.paths_columns_fused(shift=1)
    _compare(b, f, "generated")
    # and without storing the paths: same columns, lengths and errors
    g = mk()
    g.build_graph()
    g.chain_index()
    g.paths_columns_fused(shift=1, store_paths=False)
    import torch
    torch.cuda.synchronize()
    assert torch.equal(g.cols, b.cols) and torch.equal(g.path_len, b.path_len) and torch.equal(g.agent_err, b.agent_err)


def test_reset_fused_takes_the_fused_kernel_and_equals_the_dense_route():
    import torch
    fx = load_env("Test_8_Level_0")
    a = _batch(fx, 150, 50)
    b = _batch(fx, 150, 50)
    oa = a.reset_fused(fused_paths=True)
    ob = b.reset_fused(fused_paths=False)
    torch.cuda.synchronize()
    assert getattr(a, "dist", None) is None and b.dist is not None
    assert torch.equal(oa, ob) and torch.equal(a.agent_err, b.agent_err) and torch.equal(a.path_len, b.path_len)
