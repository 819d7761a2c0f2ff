"""Bookkeeping kernels (shift / delay / crash checks) against the CPU checker on seeded random planes."""
import ctypes as C

import numpy as np
import pytest

from tests.golden_util import load_env

pytestmark = pytest.mark.gpu


def _random_maps(rng, A, R, T1, tpc):
    """Planes that look like real ones: per agent a few consecutive runs on random rails."""
    m = np.zeros((A, R, T1), np.int8)
    for a in range(A):
        t = int(rng.integers(0, 3))
        while t < T1:
            ln = int(rng.integers(1, 12)) * int(tpc[a])
            r = int(rng.integers(R))
            m[a, r, t:t + ln] = rng.choice([-1, 1])
            t += ln
            if rng.random() < 0.15:
                break
    return m


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_update_and_crash_match_checker(seed):
    import flatland_marl_amd as fm
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    fx = load_env("Test_0_Level_0")
    A, depth = 24, 60
    T1 = depth + 1
    b = fm.ObservationBatch(fx["grid"], fx["initial_position"][:A], fx["direction"][:A], fx["target"][:A],
                            fx["speed"][:A], depth, max_rails=40)
    b.build_graph()
    R = int(b.R[0])
    tpc = b.tpc[0].cpu().numpy()
    vp = C.c_void_p
    for trial in range(40):
        maps = _random_maps(rng, A, R, T1, tpc)
        status = rng.integers(0, 3, size=A).astype(np.int32)
        pos = rng.integers(0, 60, size=A).astype(np.int32)
        pos[status != 1] = -1
        init = b.start_cell[0].cpu().numpy()
        b.set_agent_state(status=status, pos_cell=pos)
        a = int(rng.integers(A))
        before = int(rng.integers(2))
        nxt = int(rng.integers(0, 60))
        dmaps = b.upload_maps(maps)
        got_crash = int(b.check_crash(dmaps, [a], [before], [nxt])[0].item())
        want_crash = orc.lib().orc_check_crash(maps.ctypes.data_as(vp), A, R, T1, status.ctypes.data_as(vp),
                                               tpc.ctypes.data_as(vp), pos.ctypes.data_as(vp), init.ctypes.data_as(vp),
                                               a, before, nxt)
        assert got_crash == want_crash, (trial, a, before)
        want = maps.copy()
        orc.lib().orc_update_bitmaps(want.ctypes.data_as(vp), A, R, T1, status.ctypes.data_as(vp),
                                     tpc.ctypes.data_as(vp), a, before)
        b.update_bitmaps(dmaps, [a], [before])
        got = b.maps_numpy(0, dmaps)
        assert np.array_equal(got, want), (trial, a, before)


def _cols_of(maps):
    """[A,R,T1] planes with at most one entry per column -> [A,T1] int16 column form."""
    A, R, T1 = maps.shape
    cols = np.full((A, T1), -1, np.int16)
    a, r, t = np.nonzero(maps)
    cols[a, t] = r * 2 + (maps[a, r, t] > 0)
    return cols


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_column_form_step_matches_checker(seed):
    """flm_rollout_apply (whole per-agent loop on the column form) against the CPU checker's dense-plane
    functions driven agent by agent through the same cases, on seeded random planes, states and decisions."""
    import torch
    import flatland_marl_amd as fm
    from flatland_marl_amd.rollout import RolloutBatch
    from oracle import oracle as orc
    rng = np.random.default_rng(100 + seed)
    fx = load_env("Test_0_Level_0")
    A, depth = 24, 60
    T1 = depth + 1
    b = fm.ObservationBatch(fx["grid"], fx["initial_position"][:A], fx["direction"][:A], fx["target"][:A],
                            fx["speed"][:A], depth, max_rails=40)
    b.build_graph()
    R, W = int(b.R[0]), b.W
    tpc = b.tpc[0].cpu().numpy()
    init = b.start_cell[0].cpu().numpy()
    ro = RolloutBatch(b, fx["speed"][:A], fx["initial_direction"][:A], cap_alt=2)
    ro.cols = torch.empty((1, A, T1), dtype=torch.int16, device=ro.device)
    vp = C.c_void_p
    dev = ro.device
    for trial in range(30):
        maps = _random_maps(rng, A, R, T1, tpc)
        alt_maps = _random_maps(rng, A, R, T1, tpc)                    # one alternative plane per agent
        status = rng.choice([0, 1, 1, 1, 3], size=A).astype(np.int32)
        pos = rng.integers(0, 60, size=A).astype(np.int32)
        pos[status != 1] = -1
        # cases: done -> 0, ready -> 1, active -> 2 (before a switch, one alternative), 3 or 4
        branch = np.where(status == 3, 0, np.where(status == 0, 1, rng.integers(2, 5, size=A))).astype(np.int32)
        net_in = rng.integers(0, 2, size=A).astype(np.int32)
        has_alt = rng.random(A) < 0.8
        # paths: a few elements (cell, heading | action<<8 | next heading<<16)
        Lp = ro.Lp
        plen = rng.integers(0, 4, size=A).astype(np.int32)
        pcell = rng.integers(0, 60, size=(A, Lp)).astype(np.int32)
        pact = rng.integers(1, 5, size=(A, Lp))
        pnh = rng.integers(0, 4, size=(A, Lp))
        pmeta = (rng.integers(0, 4, size=(A, Lp)) | (pact << 8) | (pnh << 16)).astype(np.int32)
        alt_len = rng.integers(1, 4, size=A).astype(np.int32)
        alt_cell = rng.integers(0, 60, size=(A, 2, Lp)).astype(np.int32)
        alt_meta = (rng.integers(0, 4, size=(A, 2, Lp)) | (rng.integers(1, 5, size=(A, 2, Lp)) << 8)
                    | (rng.integers(0, 4, size=(A, 2, Lp)) << 16)).astype(np.int32)

        # ---- device
        t = lambda x, dt=None: torch.as_tensor(np.ascontiguousarray(x), device=dev) if dt is None else \
            torch.as_tensor(np.ascontiguousarray(x), device=dev).to(dt)
        ro.cols.copy_(t(_cols_of(maps)[None]))
        ro.status.copy_(t(status[None]))
        ro.pos.copy_(t(pos[None]))
        ro.branch.copy_(t(branch[None]))
        ro.path_cell.copy_(t(pcell[None]))
        ro.path_meta.copy_(t(pmeta[None]))
        ro.path_head.zero_()
        ro.path_len.copy_(t(plen[None]))
        ro.alt_slot.copy_(t(np.where(branch == 2, np.arange(A), -1).astype(np.int32)[None]))
        alt_cols = np.full((A, 2, T1), -1, np.int16)
        alt_cols[:, 0] = _cols_of(alt_maps)
        alt = dict(count=t(np.where(has_alt, 1, 0).astype(np.int32)), len=t(np.stack([alt_len, alt_len], 1)),
                   cell=t(alt_cell), meta=t(alt_meta), cols=t(alt_cols))
        best = torch.zeros((1, A), dtype=torch.int32, device=dev)
        ro.apply(t(net_in[None]), best, alt)
        torch.cuda.synchronize()

        # ---- checker: the same loop on dense planes (src/main.py:146-249)
        want = maps.copy()
        w_crash = np.zeros(A, np.int32)
        w_net = np.zeros(A, np.int32)
        w_act = np.zeros(A, np.int32)
        head = np.zeros(A, np.int32)
        cur_len, cur_cell, cur_meta = plen.copy(), pcell.copy(), pmeta.copy()

        def next_cell(a):
            if head[a] >= cur_len[a]:
                return -1
            c, m = int(cur_cell[a, head[a]]), int(cur_meta[a, head[a]])
            act, nh = (m >> 8) & 0xFF, (m >> 16) & 3
            return c if act == 4 else c + [-W, 1, W, -1][nh]

        def agent_action(a):
            if head[a] >= cur_len[a]:
                return 4
            act = (int(cur_meta[a, head[a]]) >> 8) & 0xFF
            head[a] += 1
            return act if 1 <= act <= 3 else 0

        def crash(a, before, nxt):
            return orc.lib().orc_check_crash(want.ctypes.data_as(vp), A, R, T1, status.ctypes.data_as(vp),
                                             tpc.ctypes.data_as(vp), pos.ctypes.data_as(vp), init.ctypes.data_as(vp),
                                             a, before, nxt)

        def update(a, before):
            orc.lib().orc_update_bitmaps(want.ctypes.data_as(vp), A, R, T1, status.ctypes.data_as(vp),
                                         tpc.ctypes.data_as(vp), a, before)

        for a in range(A):
            br, net = int(branch[a]), int(net_in[a])
            if br == 0:
                want[a] = 0
                net = 0
            elif br == 1:
                if net:
                    if crash(a, 0, -1):
                        w_crash[a], net, w_act[a] = 1, 0, 4
                    else:
                        update(a, 0)
                        w_act[a] = 2
            elif br == 2:
                if has_alt[a]:
                    want[a] = alt_maps[a]
                    cur_len[a], cur_cell[a], cur_meta[a], head[a] = alt_len[a], alt_cell[a, 0], alt_meta[a, 0], 0
                else:
                    net = 0
                if net == 0:
                    w_act[a] = 4
                elif crash(a, 1, -1):
                    w_crash[a], net, w_act[a] = 1, 0, 4
                else:
                    w_act[a] = agent_action(a)
                    update(a, 1)
            elif br == 3:
                if crash(a, 0, next_cell(a)):
                    w_crash[a], net, w_act[a] = 1, 0, 4
                else:
                    net = 1
                    w_act[a] = agent_action(a)
                    update(a, 0)
            else:
                net = 1
                update(a, 0)
            w_net[a] = net
        got = ro.maps_numpy(0, R)
        assert np.array_equal(ro.crash[0].cpu().numpy(), w_crash), trial
        assert np.array_equal(ro.net[0].cpu().numpy(), w_net), trial
        assert np.array_equal(ro.action[0].cpu().numpy(), w_act), trial
        assert np.array_equal(got, want), trial
        assert np.array_equal(ro.path_head[0].cpu().numpy(), head), trial
