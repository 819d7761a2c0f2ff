"""The package's rail generator: well-formed grids (flatland's encoding and element types, no dead ends),
agents the reference's pipeline accepts, reproducible per seed."""
import numpy as np
import pytest

from flatland_marl_amd import railgen

DR, DC = (-1, 0, 1, 0), (0, 1, 0, -1)


def _nib(code, h):
    return (int(code) >> ((3 - h) * 4)) & 15


@pytest.mark.parametrize("H,W,seed,kw", [(25, 25, 0, dict(spacing=4, max_rails=70)), (25, 25, 5, dict(spacing=3)),
                                         (30, 45, 1, dict(spacing=4, keep=0.7)), (100, 100, 2, dict(spacing=6)),
                                         (17, 60, 3, dict(spacing=2, keep=0.4))])
def test_grid_is_well_formed(H, W, seed, kw):
    g = railgen.generate_rail(H, W, seed=seed, **kw)
    assert g.shape == (H, W) and g.dtype == np.uint16
    assert np.array_equal(g, railgen.generate_rail(H, W, seed=seed, **kw))
    n_switch = 0
    for r, c in zip(*np.nonzero(g)):
        arms = set()
        for h in range(4):
            nib = _nib(g[r, c], h)
            assert bin(nib).count("1") <= 2                   # at most one choice per heading
            assert not nib & (8 >> ((h + 2) % 4))             # no U-turns: there are no dead ends
            for n in range(4):
                if nib & (8 >> n):
                    rr, cc = r + DR[n], c + DC[n]
                    assert 0 <= rr < H and 0 <= cc < W and g[rr, cc] != 0
                    assert _nib(g[rr, cc], n) != 0            # the train can go on from there
                    assert _nib(g[r, c], (n + 2) % 4) & (8 >> ((h + 2) % 4))   # and the track works both ways
                    arms.add(n)
                    arms.add((h + 2) % 4)
        assert len(arms) >= 2
        n_switch += any(bin(_nib(g[r, c], h)).count("1") == 2 for h in range(4))
    assert n_switch >= 2


def test_agents_are_acceptable_to_the_pipeline():
    from oracle import oracle as orc
    for seed in range(4):
        env = railgen.generate_env(25, 25, 10, seed=seed, spacing=4, max_rails=70)
        graph = railgen._StateGraph(env["grid"])
        for a in range(10):
            s, t = env["initial_position"][a], env["target"][a]
            assert graph.reach(s, env["direction"][a])[t[0], t[1]]
            assert graph.track[s[0], s[1]] >= 0 and graph.track[t[0], t[1]] >= 0
            assert graph.track[s[0], s[1]] != graph.track[t[0], t[1]]
        o = orc.OracleEnv(env, 150)
        assert o.g.R <= 70
        maps, paths, errs = o.get_many()
        assert not any(errs)
        assert all(p is not None for p in paths)


def test_batch_layout():
    b = railgen.make_generated_batch(40, 6, 25, 25, seed=3, n_maps=2, spacing=4, max_rails=70)
    assert b["grid"].shape == (40, 25, 25) and b["start"].shape == (40, 6, 2) and b["speed"].shape == (40, 6)
    assert len({g.tobytes() for g in b["grid"]}) > 2          # symmetries of the two maps
    for e in (0, 17, 39):
        for a in range(6):
            r, c = b["start"][e, a]
            assert b["grid"][e, r, c] != 0
            assert railgen.reachable_cells(b["grid"][e], (r, c), b["direction"][e, a])[tuple(b["target"][e, a])]
