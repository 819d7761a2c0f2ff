"""The training rollout at the size one GPU gets of the 8192-environment configuration: 1024 environments x 10
agents on generated 25x25 grids, four speeds, a malfunction process that really fires.  Whole-batch properties,
sixteen sampled environments replayed step by step against the CPU checker under the trace driver, a rerun that
must give the same bits, and a learning pass whose replay feed is compared with the script's per-agent rules
(src/main.py:269-299) for every agent of the batch.  Needs an MI355X."""
import hashlib
import types
from collections import Counter

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

E, A, DEPTH, RP, STEPS = 1024, 10, 150, 70, 64
MALF = dict(rate=40, min_duration=1, max_duration=5, seed=0)


def _data():
    from flatland_marl_amd.railgen import make_generated_batch
    return make_generated_batch(E, A, 25, 25, seed=77, n_maps=8, depth=DEPTH, spacing=4, max_rails=RP)


def _rollout(data):
    import flatland_marl_amd as fm
    from flatland_marl_amd.rollout import RolloutBatch
    b = fm.ObservationBatch(data["grid"], data["start"], data["direction"], data["target"], data["speed"], DEPTH,
                            max_rails=RP)
    seeds = (np.arange(E, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(12345)) % np.uint64(2 ** 32)
    ro = RolloutBatch(b, data["speed"], data["direction"], max_alt_depth=202, malfunction=dict(MALF, env_seeds=seeds))
    return ro.reset(), seeds


def _checker_trace(data, e, seed):
    from oracle.oracle_builder import OracleRailObs, OraclePreprocessor
    from oracle.trace_driver import run_trace
    from tests.test_oracle_rollout import shim_env
    fx = dict(grid=data["grid"][e], initial_position=data["start"][e], initial_direction=data["direction"][e],
              direction=data["direction"][e], target=data["target"][e], speed=data["speed"][e],
              handle=np.arange(A, dtype=np.int32), status=np.zeros(A, np.int32))
    ob = OracleRailObs(DEPTH)
    env = shim_env(fx, ob, A, dict(MALF, seed=int(seed)))
    try:
        rec, _ = run_trace(env, ob, OraclePreprocessor(RP), STEPS, hash_obs=False)
    except (IndexError, KeyError, AssertionError) as ex:
        # the reference's builder raises for this environment (an alternative that ends on its target before any
        # switch, src/rail_observations.py:348-350): the script would stop there; the batch flags the agent instead
        return type(ex).__name__
    return rec


def _state_digest(ro):
    import torch
    h = hashlib.sha256()
    for t in (ro.cols, ro.status, ro.pos, ro.heading, ro.path_head, ro.path_len, ro.malfunction, ro.done):
        h.update(t.cpu().numpy().tobytes())
    return h.hexdigest()


def test_rollout_1024_environments_with_malfunctions():
    import torch
    from flatland_marl_amd.rollout import hash_policy
    data = _data()
    ro, seeds = _rollout(data)
    b = ro.b
    sample = sorted(np.random.default_rng(4).choice(E, size=28, replace=False).tolist())
    traces = {e: _checker_trace(data, e, seeds[e]) for e in sample}
    raising = [e for e in sample if isinstance(traces[e], str)]
    sample = [e for e in sample if e not in raising]
    assert len(sample) >= 16
    rail = torch.as_tensor(data["grid"].reshape(E, -1) != 0, device=ro.device)
    R = torch.as_tensor(b.R.astype(np.int64), device=ro.device)
    fired = 0
    digests = []
    for rerun in range(2):
        if rerun:
            ro.reset()
        for t in range(STEPS):
            info = ro.step(hash_policy)
            # ---- every environment: legal states
            st, pos = ro.status, ro.pos
            assert int(st.min()) >= 0 and int(st.max()) <= 3
            on_grid = pos >= 0
            assert bool((on_grid == (st == 1)).all())                       # exactly the active agents stand on the grid
            assert bool(rail.gather(1, pos.clamp(min=0).long())[on_grid].all())   # and they stand on rail cells
            key = torch.where(on_grid, pos + 1, -torch.arange(1, A + 1, device=ro.device, dtype=pos.dtype)[None, :])
            assert all(len(torch.unique(key[e])) == A for e in range(0, E, 97))   # no two agents in one cell
            srt = torch.sort(key, dim=1).values
            assert bool((srt[:, 1:] != srt[:, :-1]).all())
            assert int(ro.action.min()) >= 0 and int(ro.action.max()) <= 4
            cols = ro.cols
            assert int(cols.min()) >= -1 and bool((cols.long() < (2 * R)[:, None, None]).all())
            fired += int((ro.malfunction > 0).sum().item())
            if rerun:
                continue
            # ---- the sampled environments against the checker's trace, step by step
            arrs = {k: v.cpu().numpy() for k, v in (("branch", ro.branch), ("net", ro.net), ("crash", ro.crash),
                                                     ("action", ro.action), ("status", ro.status), ("pos", ro.pos),
                                                     ("malf", ro.malfunction))}
            slot = ro.alt_slot.cpu().numpy()
            cnt = info["alt"]["count"].cpu().numpy() if info["alt"] is not None else None
            best = info["best"].cpu().numpy()
            for e in sample:
                g = traces[e]
                if t >= int(g["n_steps_run"]):
                    continue
                for k, v in arrs.items():
                    assert np.array_equal(v[e], g[k][t]), (e, t, k, v[e], g[k][t])
                n_alt = np.zeros(A, np.int32)
                bst = np.full(A, -1, np.int32)
                for a in np.nonzero(slot[e] >= 0)[0]:
                    n_alt[a] = cnt[slot[e, a]]
                    if n_alt[a] > 0:
                        bst[a] = best[e, a]
                assert np.array_equal(n_alt, g["n_alt"][t]) and np.array_equal(bst, g["best"][t]), (e, t)
                maps = ro.maps_numpy(e)
                assert hashlib.sha256(np.ascontiguousarray(maps).tobytes()).hexdigest() == str(g["maps_sha"][t]), (e, t)
        torch.cuda.synchronize()
        digests.append(_state_digest(ro))
    assert fired > 1000                                                      # the malfunction process is not idle
    assert any(int(traces[e]["malf"].max()) > 0 for e in sample)
    assert digests[0] == digests[1]                                          # the same launches give the same bits
    # flags: none in the environments the checker played through, one at least where the reference raises, and
    # never because a capacity of the batch was too small
    err = ro.agent_err.cpu().numpy()
    assert not err[sample].any()
    assert all(err[e].any() for e in raising)
    assert ro.flagged_agents()[1] == 0


def _row_hash(t, w):
    import torch
    return (t.reshape(t.shape[0], -1).to(torch.int64) * w).sum(dim=1)


def test_learning_pass_1024_environments_feeds_the_replay_as_the_script_does():
    import torch
    from flatland_marl_amd.agent import DQNAgent
    from flatland_marl_amd.train import BatchedTrainer
    data = _data()
    ro, _ = _rollout(data)
    b = ro.b
    args = types.SimpleNamespace(prediction_depth=DEPTH, lr=5e-5, buffer_size=40000, batch_size=512, update_every=10,
                                 gamma=0.99, tau=1e-3)
    torch.manual_seed(0)
    agent = DQNAgent(args, action_space=2, bitmap_height=3 * RP)
    tr = BatchedTrainer(ro, agent, start_eps=0.3, crash_penalty=True, max_learns_per_step=2, seed=3)
    w = torch.randint(-2 ** 40, 2 ** 40, (3 * b.plane_stride,), dtype=torch.int64, device=ro.device,
                      generator=torch.Generator(device=ro.device).manual_seed(1))
    rp = tr.replay

    n = E * A
    curr = np.zeros(n, np.int64)                      # hash of the observation each agent last acted on (curr_obs)
    has_curr = np.zeros(n, bool)
    was_done = np.zeros(n, bool)
    total = 0
    before = agent.qnetwork_local.flat_params().clone()
    for t in range(20):
        head = rp.head
        k = tr.step(train=True)
        # what the step wrote into the ring
        sl = (head + torch.arange(k, device=ro.device)) % rp.capacity
        assert k <= rp.capacity
        stored = list(zip(_row_hash(rp.state[sl], w).cpu().numpy().tolist(), rp.action[sl].cpu().numpy().tolist(),
                          rp.reward[sl].cpu().numpy().tolist(), _row_hash(rp.next_state[sl], w).cpu().numpy().tolist(),
                          (rp.done[sl].cpu().numpy() != 0).tolist()))
        update = np.zeros(n, bool)
        for req, obs, _ in (tr.policy.last or []):
            r = req.cpu().numpy()
            update[r] = True
            curr[r] = _row_hash(obs, w).cpu().numpy()
            has_curr[r] = True
        net = ro.net.view(-1).cpu().numpy()
        crash = ro.crash.view(-1).cpu().numpy() != 0
        reward = ro.reward.view(-1).to(torch.float32).cpu().numpy()
        done = ro.done.view(-1).cpu().numpy() != 0
        want = []
        pen = np.nonzero(crash & has_curr)[0]                                   # src/main.py:271-273
        want.extend(zip(curr[pen].tolist(), [1] * len(pen), [-100.0] * len(pen), curr[pen].tolist(), [True] * len(pen)))
        upd = np.nonzero(update & ~was_done)[0]                                 # src/main.py:275-278
        if len(upd):
            nxt = _row_hash(ro.consult_observations(torch.as_tensor(upd.astype(np.int32), device=ro.device)), w).cpu().numpy()
            want.extend(zip(curr[upd].tolist(), net[upd].tolist(), reward[upd].tolist(), nxt.tolist(), done[upd].tolist()))
        was_done = done
        assert Counter(stored) == Counter(want), t
        total += len(want)
    torch.cuda.synchronize()
    assert total > 10000 and tr.n_learn > 0
    assert torch.isfinite(agent.last_loss).all()
    after = agent.qnetwork_local.flat_params()
    assert torch.isfinite(after).all() and not torch.equal(before, after)
    assert ro.flagged_agents()[1] == 0


def test_alternative_capacity_overflow_surfaces():
    """More alternatives than the rollout was sized for is not a case of the reference: the agents concerned are
    flagged, counted, and the episode summary raises instead of training on agents that stand still."""
    import torch
    import flatland_marl_amd as fm
    from flatland_marl_amd.agent import DQNAgent
    from flatland_marl_amd.rollout import RolloutBatch
    from flatland_marl_amd.train import BatchedTrainer
    from tests.golden_util import load_env
    fx = load_env("Test_2_Level_0")
    A_, depth, Rp = 12, 40, 70
    rep = lambda x: np.asarray(x)[None, :A_]
    b = fm.ObservationBatch(fx["grid"][None], rep(fx["initial_position"]), rep(fx["initial_direction"]), rep(fx["target"]),
                            rep(fx["speed"]), depth, max_rails=Rp)
    with pytest.raises(ValueError):
        RolloutBatch(b, rep(fx["speed"]), rep(fx["initial_direction"]), max_alt_depth=0)      # paths would not fit
    ro = RolloutBatch(b, rep(fx["speed"]), rep(fx["initial_direction"]), cap_alt=1).reset()
    args = types.SimpleNamespace(prediction_depth=depth, lr=1e-3, buffer_size=500, batch_size=16, update_every=4,
                                 gamma=0.99, tau=1e-2)
    torch.manual_seed(0)
    tr = BatchedTrainer(ro, DQNAgent(args, action_space=2, bitmap_height=3 * Rp), start_eps=1.0, end_eps=1.0, max_steps=120)
    with pytest.raises(OverflowError):
        tr.run_episode(train=False)
    assert ro.flagged_agents()[1] > 0
