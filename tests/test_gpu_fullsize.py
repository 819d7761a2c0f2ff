"""The three synthetic workloads at their full sizes (4096x50 on 100x100, 256x200 on 150x150 at depth 300,
1024x10 on 35x35): size-independent properties over the whole batch, and sampled environments against the
CPU checker.  Needs an MI355X (the 4096-environment batch holds ~55 GB of observations)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _workload(name):
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from bench import workload_batch
    return workload_batch(name)[:4]


def _checksum(t):
    import torch
    flat = t.reshape(-1)
    n8 = flat.numel() // 8 * 8
    return int(flat[:n8].view(torch.int64).sum().item()) + int(flat[n8:].sum(dtype=torch.int64).item())


@pytest.mark.parametrize("name,n_sample", [("cfg3", 12), ("cfg3g", 6), ("cfg5", 2), ("cfg2", 24), ("cfg2g", 24)])
def test_full_size_workload(name, n_sample):
    import torch
    import flatland_marl_amd as fm
    from oracle import oracle as orc
    data, E, A, depth = _workload(name)
    b = fm.ObservationBatch(data["grid"], data["start"], data["direction"], data["target"], data["speed"], depth)
    obs = b.reset_fused()
    torch.cuda.synchronize()
    b.check_graph()
    Rp, T1, ps = b.Rp, b.T1, b.plane_stride
    err = b.agent_err.cpu().numpy()
    assert obs.shape == (E, A, 3, ps)

    # 1. heat + own is the same total for every agent of an environment; heat sums are (A-1) x own sums;
    #    every own plane has at most one entry per time step and only values -1, 0, +1
    step = max(1, (1 << 30) // (A * 3 * ps))
    for lo in range(0, E, step):
        o = obs[lo:lo + step]
        own = o[:, :, 0].view(torch.int8)
        pos_own = (own > 0).to(torch.uint8)
        neg_own = (own < 0).to(torch.uint8)
        assert bool(((own >= -1) & (own <= 1)).all())
        tot_p = o[:, :, 1] + pos_own
        tot_n = o[:, :, 2] + neg_own
        assert bool((tot_p == tot_p[:, :1]).all()) and bool((tot_n == tot_n[:, :1]).all())
        per_col = (own[:, :, :Rp * T1] != 0).reshape(-1, A, Rp, T1).sum(dim=2)
        assert int(per_col.max().item()) <= 1
        assert not bool(o[:, :, :, Rp * T1:].any())
        del o, own, pos_own, neg_own, tot_p, tot_n, per_col

    # 2. the same launch sequence again gives the same bits
    c0 = _checksum(obs)
    b.reset_fused()
    torch.cuda.synchronize()
    assert _checksum(obs) == c0

    # 2b. the per-request kernel (rollouts) writes the same bytes as the sweep, also on planes of many tiles
    from flatland_marl_amd._lib import call, ptr
    req = torch.as_tensor(np.random.default_rng(5).choice(E * A, size=min(97, E * A), replace=False).astype(np.int32),
                          device=obs.device)
    one = torch.empty((req.numel(), 3, ps), dtype=torch.uint8, device=obs.device)
    call("flm_observations_req_columns", ptr(b.cols), ptr(req), None, None, req.numel(), E, A, Rp, T1, ps, ptr(one),
         b.stream.cuda_stream)
    assert torch.equal(one, obs.view(E * A, 3, ps)[req.long()])
    del one

    # 3. sampled environments against the CPU checker: planes, heat maps, paths, errors
    rng = np.random.default_rng(7)
    names = {"": 0, "IndexError": 2, "KeyError": 3, "AssertionError": 4}
    for e in sorted(rng.choice(E, size=n_sample, replace=False)):
        fx = dict(grid=data["grid"][e], initial_position=data["start"][e], direction=data["direction"][e],
                  target=data["target"][e], speed=data["speed"][e], handle=np.arange(A))
        env = orc.OracleEnv(fx, depth)
        maps, paths, errs = env.get_many()
        assert [names[x] for x in errs] == list(err[e]), e
        R = env.g.R
        assert int(b.R[e]) == R
        got = obs[e, :, :, :Rp * T1].cpu().numpy().reshape(A, 3, Rp, T1)
        ok = [a for a in range(A) if not errs[a]]
        assert np.array_equal(got[ok, 0, :R].view(np.int8), maps[ok]), e
        assert not got[:, :, R:].any()
        tot_p = (maps > 0).sum(axis=0)
        tot_n = (maps < 0).sum(axis=0)
        for a in ok[:: max(1, len(ok) // 8)]:
            assert np.array_equal(got[a, 1, :R], tot_p - (maps[a] > 0)), (e, a)
            assert np.array_equal(got[a, 2, :R], tot_n - (maps[a] < 0)), (e, a)
        dev_paths = b.paths_numpy(int(e))
        for a in range(A):
            if paths[a] is None:
                assert dev_paths[a] is None
            else:
                assert np.array_equal(dev_paths[a], paths[a]), (e, a)
    del obs, b
    torch.cuda.empty_cache()
