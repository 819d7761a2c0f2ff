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
