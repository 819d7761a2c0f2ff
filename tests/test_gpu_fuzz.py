"""Randomly generated environments (the package's own rail generator: slips, crossings, switches two cells
apart, agents the reference would reject) through the whole path against the CPU checker, which itself is pinned
by the reference's goldens (needs an MI355X)."""
import numpy as np
import pytest

from flatland_marl_amd import railgen

pytestmark = pytest.mark.gpu
ERR = {"": 0, "IndexError": 2, "KeyError": 3, "AssertionError": 4}

# (H, W, agents, depth, environments, generator arguments, targets on the start's own track allowed)
CASES = [
    (12, 12, 3, 30, 24, dict(spacing=2, keep=0.6), False),
    (20, 31, 7, 60, 24, dict(spacing=3, keep=0.5), True),
    (25, 25, 10, 150, 32, dict(spacing=4, max_rails=70), False),
    (33, 18, 17, 97, 16, dict(spacing=2, keep=0.45), True),
    (48, 48, 33, 200, 12, dict(spacing=5, keep=0.7), False),
    (64, 40, 64, 120, 8, dict(spacing=3, keep=0.4), True),
]


def _batch(H, W, A, depth, E, kw, same_track, seed):
    envs = [railgen.generate_env(H, W, A, seed=seed * 1000 + e, allow_same_track=same_track, depth=depth, **kw)
            for e in range(E)]
    stack = lambda k: np.stack([v[k] for v in envs])
    return envs, dict(grid=stack("grid"), start=stack("initial_position"), direction=stack("direction"),
                      target=stack("target"), speed=stack("speed"))


@pytest.mark.parametrize("case", range(len(CASES)))
def test_generated_environments_against_checker(case):
    import torch
    import flatland_marl_amd as fm
    from oracle import oracle as orc
    H, W, A, depth, E, kw, same_track = CASES[case]
    envs, data = _batch(H, W, A, depth, E, kw, same_track, seed=case + 1)
    b = fm.ObservationBatch(data["grid"], data["start"], data["direction"], data["target"], data["speed"], depth)
    obs = b.reset_fused()
    torch.cuda.synchronize()
    b.check_graph()
    Rp, T1 = b.Rp, b.T1
    err = b.agent_err.cpu().numpy()
    n_flagged = 0
    for e in range(E):
        env = orc.OracleEnv(envs[e], depth)
        maps, paths, errs = env.get_many()
        assert [ERR[x] for x in errs] == list(err[e]), (case, e)
        n_flagged += sum(1 for x in errs if x)
        R = env.g.R
        assert int(b.R[e]) == R
        assert np.array_equal(b.edge_info[e, :R].cpu().numpy(), env.g.edge_info)
        got = obs[e, :, :, :Rp * T1].cpu().numpy().reshape(A, 3, Rp, T1)
        ok = [a for a in range(A) if not errs[a]]
        assert np.array_equal(got[ok, 0, :R].view(np.int8), maps[ok]), (case, e)
        assert not got[:, :, R:].any()
        tot_p = (maps > 0).sum(axis=0)
        tot_n = (maps < 0).sum(axis=0)
        for a in ok:
            assert np.array_equal(got[a, 1, :R], tot_p - (maps[a] > 0)), (case, e, a)
            assert np.array_equal(got[a, 2, :R], tot_n - (maps[a] < 0)), (case, e, a)
        dev_paths = b.paths_numpy(e)
        for a in range(A):
            if paths[a] is None:
                assert dev_paths[a] is None
            else:
                assert np.array_equal(dev_paths[a], paths[a]), (case, e, a)
    if not same_track:
        assert n_flagged == 0


@pytest.mark.parametrize("H,W,A,depth,steps,kw,seed", [
    (25, 25, 10, 80, 120, dict(spacing=4, max_rails=70), 11),
    (30, 44, 24, 150, 150, dict(spacing=3, keep=0.5), 12),
    (21, 21, 12, 40, 100, dict(spacing=2, keep=0.55), 13),
    # 44 agents x 401 columns: more than the 32 KiB of columns the per-agent walk stages in LDS, so this case
    # takes the walk that leaves them in global memory
    (30, 44, 44, 400, 120, dict(spacing=3, keep=0.5), 17),
])
def test_generated_rollout_against_checker(H, W, A, depth, steps, kw, seed):
    """A full rollout on a generated grid: the trace the CPU checker's builder produces under the trace driver
    is the reference here; the batched driver must reproduce it (with malfunctions on)."""
    from oracle.oracle_builder import OracleRailObs, OraclePreprocessor
    from oracle.trace_driver import run_trace
    from tests.test_oracle_rollout import shim_env
    from tests.test_gpu_rollout_batched import replay_against
    fx = railgen.generate_env(H, W, A, seed=seed, depth=depth, **kw)
    malf = dict(rate=30, min_duration=0, max_duration=3, seed=seed)
    ob = OracleRailObs(depth)
    env = shim_env(fx, ob, A, malf)
    from oracle import oracle as orc
    Rp = max(100, orc.OracleEnv(fx, depth).g.R)
    rec, snaps = run_trace(env, ob, OraclePreprocessor(Rp), steps)
    rec.update(agents=np.array(A), depth=np.array(depth), steps=np.array(steps), max_rails=np.array(Rp),
               malfunction=np.array([malf["rate"], malf["min_duration"], malf["max_duration"], malf["seed"]]))
    replay_against(fx, rec)
