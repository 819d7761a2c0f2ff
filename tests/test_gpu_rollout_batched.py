"""The batched, device-resident rollout against the reference's recorded traces (needs an MI355X).

Every golden trace was recorded by driving the reference's own observation builder through
src/main.py's five per-agent cases with a hash in place of the network (oracle/trace_driver.py).  Here
the batched driver plays the same decisions for several copies of the environment at once and must
reproduce every recorded case, decision, crash flag, action, chosen alternative, agent state, occupancy
map and consulted observation."""
import glob
import hashlib
import os

import numpy as np
import pytest

from tests.golden_util import GOLDEN, load_env, dense_from_runs
from tests.test_oracle_rollout import malfunction_of

pytestmark = pytest.mark.gpu
TRACES = sorted(glob.glob(os.path.join(GOLDEN, "rollout", "*.npz")))


def _rollout(fx, A, depth, max_rails, E, malfunction=None):
    import flatland_marl_amd as fm
    from flatland_marl_amd.rollout import RolloutBatch
    rep = lambda x: np.repeat(np.asarray(x)[None, :A], E, axis=0)
    b = fm.ObservationBatch(np.repeat(fx["grid"][None], E, axis=0), rep(fx["initial_position"]),
                            rep(fx["initial_direction"]), rep(fx["target"]), rep(fx["speed"]), depth, max_rails=max_rails)
    if malfunction is not None:
        # every copy draws from the recorded environment's seed so that the copies stay in lockstep
        malfunction = dict(malfunction, env_seeds=[malfunction["seed"]] * E)
    return RolloutBatch(b, rep(fx["speed"]), rep(fx["initial_direction"]), malfunction=malfunction).reset()


@pytest.mark.parametrize("path", TRACES, ids=[os.path.basename(p)[:-4] for p in TRACES])
def test_batched_rollout_replays_trace(path):
    gold = dict(np.load(path))
    name = os.path.basename(path).split("_a")[0]
    replay_against(load_env(name), gold)


def replay_against(fx, gold):
    """Drive the batched rollout with the trace driver's decisions and compare every recorded quantity."""
    import torch
    from flatland_marl_amd.rollout import hash_policy
    A, depth, steps, Rp = int(gold["agents"]), int(gold["depth"]), int(gold["n_steps_run"]), int(gold["max_rails"])
    E = 3
    ro = _rollout(fx, A, depth, Rp, E, malfunction_of(gold))
    b = ro.b
    T1 = b.T1
    R = int(b.R[0])
    obs_h = hashlib.sha256()

    def policy(rollout, info):
        # hash the observations the reference computes for the consulting agents of environment 0, in agent order
        br = info["branch"][0].cpu().numpy()
        alt = info["alt"]
        reqs, own = [], []
        slot = rollout.alt_slot[0].cpu().numpy()
        cnt = alt["count"].cpu().numpy() if alt is not None else None
        for a in range(A):
            if br[a] == 1:
                reqs.append(a)
                own.append(rollout.cols[0, a])
            elif br[a] == 2:
                for i in range(int(cnt[slot[a]])):
                    reqs.append(a)
                    own.append(alt["cols"][slot[a], i])
        if reqs:
            req = torch.as_tensor(np.array(reqs, np.int32), device=rollout.device)
            o = rollout.consult_observations(req, torch.stack(own).contiguous())
            o = o[:, :, :Rp * T1].cpu().numpy().reshape(len(reqs), 3, Rp, T1)
            for k in range(len(reqs)):
                x = o[k].astype(np.int16)
                x[0] = o[k, 0].view(np.int8)
                obs_h.update(x.tobytes())
        return hash_policy(rollout, info)

    for t in range(steps):
        info = ro.step(policy)
        torch.cuda.synchronize()
        branch = ro.branch.cpu().numpy()
        assert (branch == branch[0]).all(), t            # the copies stay in lockstep
        assert np.array_equal(branch[0], gold["branch"][t]), t
        n_alt = np.zeros(A, np.int32)
        best = np.full(A, -1, np.int32)
        if info["alt"] is not None:
            slot = ro.alt_slot[0].cpu().numpy()
            cnt = info["alt"]["count"].cpu().numpy()
            for a in np.nonzero(slot >= 0)[0]:
                n_alt[a] = cnt[slot[a]]
                if n_alt[a] > 0:
                    best[a] = info["best"][0, a].item()
        assert np.array_equal(n_alt, gold["n_alt"][t]), t
        assert np.array_equal(best, gold["best"][t]), t
        for k, got in (("net", ro.net), ("crash", ro.crash), ("action", ro.action), ("status", ro.status), ("pos", ro.pos)):
            g = got.cpu().numpy()
            assert (g == g[0]).all(), (k, t)
            assert np.array_equal(g[0], gold[k][t]), (k, t, g[0], gold[k][t])
        if ro.malfunction is not None:
            g = ro.malfunction.cpu().numpy()
            assert (g == g[0]).all() and np.array_equal(g[0], gold["malf"][t]), t
        maps = ro.maps_numpy(0, R)
        assert hashlib.sha256(np.ascontiguousarray(maps).tobytes()).hexdigest() == str(gold["maps_sha"][t]), t
        assert np.array_equal(ro.maps_numpy(E - 1, R), maps)
        if "snap_%d" % t in gold:
            assert np.array_equal(maps, dense_from_runs(gold["snap_%d" % t], maps.shape)), t
    assert obs_h.hexdigest() == str(gold["obs_sha"])
    assert not ro.agent_err.any()


def test_counted_launches_match_the_sized_ones():
    """flm_altpaths_counted / flm_observations_req_columns_counted (sized by an upper bound, the real number read on the
    device) against flm_altpaths / flm_observations_req_columns on the same requests some steps into an episode: the
    same bits for the requests that exist, nothing written for the rest, the per-path agent list as the host would
    build it; and a launch that covers only the tail of the requests (live_base > 0)."""
    import torch
    from flatland_marl_amd._lib import call, ptr
    from flatland_marl_amd.rollout import hash_policy
    fx = load_env("Test_3_Level_0")
    A = 14
    ro = _rollout(fx, A, 60, 160, 5)
    b = ro.b
    for _ in range(25):
        ro.step(hash_policy)
    ro.classify()
    call("flm_rollout_requests", ptr(ro.branch), ptr(ro.status), ptr(ro.pos), ptr(ro.heading), ptr(b.start_cell), ro.E, ro.A,
         ptr(ro.ready_list), ptr(ro.switch_list), ptr(ro.req_counts), ptr(ro.alt_slot), ptr(ro._net), ptr(ro._best),
         ptr(b.pos_cell), ptr(b.direction), ptr(b.status), ptr(ro._req_ws), ro.stream)
    n_ready, n = ro.read_counts()[:2]
    assert n >= 4
    req = ro.switch_list[:n].clone()
    K, Lp, dev = ro.cap_alt, ro.Lp, ro.device
    n_max = n + 7
    i32 = dict(dtype=torch.int32, device=dev)

    def tables():
        return dict(count=torch.full((n_max,), -7, **i32), len=torch.full((n_max, K), -7, **i32),
                    cell=torch.full((n_max, K, Lp), -7, **i32), meta=torch.full((n_max, K, Lp), -7, **i32),
                    err=torch.full((n_max,), -7, **i32))
    fixed, state = ro._alt_tables()
    tail = (ro.E, ro.A, b.H, b.W, b.cap_cells, ro.max_alt_depth, K, Lp)
    want, got, got2 = tables(), tables(), tables()
    call("flm_altpaths", *fixed, ptr(req), n, *state, *tail, ptr(want["count"]), ptr(want["len"]), ptr(want["cell"]),
         ptr(want["meta"]), ptr(want["err"]), ro.stream)
    live = torch.tensor([n], **i32)
    rep = torch.full((n_max, K), -7, **i32)
    req_pad = torch.cat([req, torch.full((7,), 10 ** 6, **i32)])            # what lies beyond the list must not be read
    call("flm_altpaths_counted", *fixed, ptr(req_pad), n_max, ptr(live), 0, ptr(rep), *state, *tail, ptr(got["count"]),
         ptr(got["len"]), ptr(got["cell"]), ptr(got["meta"]), ptr(got["err"]), ro.stream)
    # the tail alone: requests lo.. of the n, through pointers that start at request lo
    lo = n // 2
    call("flm_altpaths_counted", *fixed, req_pad.data_ptr() + 4 * lo, n_max - lo, ptr(live), lo, None, *state, *tail,
         got2["count"].data_ptr() + 4 * lo, got2["len"].data_ptr() + 4 * K * lo, got2["cell"].data_ptr() + 4 * K * Lp * lo,
         got2["meta"].data_ptr() + 4 * K * Lp * lo, got2["err"].data_ptr() + 4 * lo, ro.stream)
    torch.cuda.synchronize()
    cnt = want["count"][:n]
    assert int(cnt.max()) >= 2
    for k in ("count", "err"):
        assert torch.equal(got[k][:n], want[k][:n]) and bool((got[k][n:] == -7).all())
        assert torch.equal(got2[k][lo:n], want[k][lo:n]) and bool((got2[k][:lo] == -7).all()) and bool((got2[k][n:] == -7).all())
    assert torch.equal(got["len"][:n], want["len"][:n]) and torch.equal(got2["len"][lo:n], want["len"][lo:n])
    used = torch.arange(Lp, device=dev)[None, None, :] < want["len"][:n, :, None]     # the elements of existing paths
    for k in ("cell", "meta"):
        assert torch.equal(got[k][:n][used], want[k][:n][used]) and bool((got[k][n:] == -7).all())
        assert torch.equal(got2[k][lo:n][used[lo:]], want[k][lo:n][used[lo:]])
    assert torch.equal(rep[:n], req[:, None].expand(n, K)) and bool((rep[n:] == -7).all())
    # observations: the departing agents' own planes, sized by an upper bound
    m = n_ready + n
    agents = torch.cat([ro.ready_list[:n_ready], req])
    want_o = ro.consult_observations(agents)
    got_o = torch.full((m + 5, 3, b.plane_stride), 9, dtype=torch.uint8, device=dev)
    live_o = torch.tensor([m], **i32)
    agents_pad = torch.cat([agents, torch.full((5,), 10 ** 6, **i32)])
    call("flm_observations_req_columns_counted", ptr(ro.cols), ptr(agents_pad), None, None, m + 5, ptr(live_o), 0, ro.E, ro.A,
         b.Rp, b.T1, b.plane_stride, ptr(got_o), ro.stream)
    torch.cuda.synchronize()
    assert torch.equal(got_o[:m], want_o) and bool((got_o[m:] == 9).all()) and int(want_o.max()) > 0
    # ... and with only the rails the environments have written: the same bytes there, the rest left alone
    limit = ro.plane_bytes_used()
    assert 0 < limit < b.plane_stride and limit % 16 == 0 and limit >= ro.rails_used() * b.T1
    assert not bool(want_o[:, :, limit:].any())                   # what lies behind is zero in the full observation
    got_l = torch.full((m + 5, 3, b.plane_stride), 9, dtype=torch.uint8, device=dev)
    call("flm_observations_req_columns_counted", ptr(ro.cols), ptr(agents_pad), None, None, m + 5, ptr(live_o), limit, ro.E,
         ro.A, b.Rp, b.T1, b.plane_stride, ptr(got_l), ro.stream)
    torch.cuda.synchronize()
    assert torch.equal(got_l[:m, :, :limit], want_o[:, :, :limit]) and bool((got_l[:m, :, limit:] == 9).all())
    assert bool((got_l[m:] == 9).all())
