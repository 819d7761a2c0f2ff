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
