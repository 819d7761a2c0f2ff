"""Batched training driver: plumbing between the rollout, the replay memory and the DQN kernels (needs an MI355X)."""
import types

import numpy as np
import pytest

from tests.golden_util import load_env

pytestmark = pytest.mark.gpu


def _setup(E=4, A=10, depth=150, Rp=100, name="Test_0_Level_0", **kw):
    import torch
    import flatland_marl_amd as fm
    from flatland_marl_amd.agent import DQNAgent
    from flatland_marl_amd.rollout import RolloutBatch
    from flatland_marl_amd.train import BatchedTrainer
    fx = load_env(name)
    rep = lambda x: np.repeat(np.asarray(x)[None, :A], E, axis=0)
    b = fm.ObservationBatch(np.repeat(fx["grid"][None], E, axis=0), rep(fx["initial_position"]),
                            rep(fx["initial_direction"]), rep(fx["target"]), rep(fx["speed"]), depth, max_rails=Rp)
    ro = RolloutBatch(b, rep(fx["speed"]), rep(fx["initial_direction"])).reset()
    args = types.SimpleNamespace(prediction_depth=depth, lr=5e-5, buffer_size=2000, batch_size=32, update_every=10,
                                 gamma=0.99, tau=1e-3)
    torch.manual_seed(0)
    agent = DQNAgent(args, action_space=2, bitmap_height=3 * Rp)
    return ro, agent, BatchedTrainer(ro, agent, **kw)


def test_policy_q_values_match_act():
    import torch
    ro, agent, tr = _setup()
    b = ro.b
    ro.classify()
    req = torch.nonzero(ro.branch.view(-1) == 1).view(-1).to(torch.int32)
    assert req.numel() == ro.E * ro.A                       # every agent is ready to depart after reset
    obs = ro.consult_observations(req)
    q = tr.policy.q_values(ro, obs).cpu().numpy()
    o = obs[:, :, :b.Rp * b.T1].cpu().numpy().reshape(-1, 3, b.Rp, b.T1)
    for k in (0, 7, req.numel() - 1):
        state = o[k].astype(np.float64)
        state[0] = o[k, 0].view(np.int8)
        want = agent.act(state.reshape(3 * b.Rp, b.T1)).cpu().numpy()
        assert np.allclose(q[k], want, rtol=1e-5, atol=1e-6)
    # the byte-input forward equals the float-input forward of the same observations
    from flatland_marl_amd.train import obs_to_f32
    qf = agent.qnetwork_local(obs_to_f32(obs, b.Rp, b.T1, b.plane_stride)).cpu().numpy()
    assert np.allclose(q, qf, rtol=1e-5, atol=1e-6)
    for n in (1, 3, 257):                                    # ragged row tiles and the small-batch dense layer
        qa = agent.qnetwork_local.forward_obs(obs[:n], b.plane_stride).cpu().numpy()
        assert np.allclose(qa, qf[:n], rtol=1e-5, atol=1e-6)
    # the observation of a request equals the all-agents sweep
    full = b.observations_fused()
    assert torch.equal(full.view(-1, 3, b.plane_stride)[req.long()], obs)


def test_training_episode_runs_and_learns():
    import torch
    ro, agent, tr = _setup(start_eps=0.5, crash_penalty=True, max_learns_per_step=2, max_steps=60)
    before = agent.qnetwork_local.flat_params().clone()
    out = tr.run_episode(train=True)
    torch.cuda.synchronize()
    assert out["steps"] >= 1 and out["transitions"] > 0 and len(tr.replay) > 0
    assert out["learns"] > 0
    assert torch.isfinite(agent.last_loss).all()
    after = agent.qnetwork_local.flat_params()
    assert torch.isfinite(after).all() and not torch.equal(before, after)
    assert int(ro.action.min()) >= 0 and int(ro.action.max()) <= 4
    st = ro.status.cpu().numpy()
    assert set(np.unique(st)).issubset({0, 1, 3})
    assert not ro.agent_err.any()
    # stored transitions are observations: own plane in {-1,0,1}, rewards <= 0 or the crash penalty
    n = len(tr.replay)
    own = tr.replay.state[:n, 0].view(torch.int8)
    assert int(own.min()) >= -1 and int(own.max()) <= 1
    r = tr.replay.reward[:n]
    assert bool(((r <= 0) | (r == -100)).all())
    out2 = tr.run_episode(train=True)                         # a second episode resets cleanly
    assert out2["steps"] >= 1 and out2["eps"] < 0.5


def test_training_with_the_optimizer_in_its_compact_form_gives_the_same_weights():
    """The trainer's learn steps run over the used rails with the optimizer's data of the first dense layer kept
    without the empty rows (packed gradient, packed moments, one pair of moments per unit and filter for the rows that
    are empty).  The same episode with that form switched off (every moment kept, gradient scattered into the flat
    layout) must leave the same parameters, target parameters and moments, bit for bit."""
    import torch
    out = []
    for compact in (True, False):
        ro, agent, tr = _setup(E=6, A=8, depth=40, Rp=40, max_learns_per_step=3, start_eps=0.3, max_steps=30, seed=3)
        agent.args.lr = 1e-3
        agent.optimizer.lr = 1e-3
        if not compact:
            agent.optimizer._ensure()
            agent.optimizer._uniform_from = None               # as after any step of another kind: the form is never taken
        summary = tr.run_episode(train=True)
        torch.cuda.synchronize()
        assert summary["learns"] >= 10
        used_form = agent.optimizer._compact is not None
        assert used_form == compact
        out.append((agent.qnetwork_local.flat_params().clone(), agent.qnetwork_target.flat_params().clone(),
                    agent.optimizer.m.clone(), agent.optimizer.v.clone(), summary["learns"], summary["transitions"]))
    assert out[0][4:] == out[1][4:]
    for x, y in zip(out[0][:4], out[1][:4]):
        assert torch.equal(x, y)
    assert out[0][2].abs().max() > 0


def _hash(t):
    import hashlib
    return hashlib.sha1(t.contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


@pytest.mark.parametrize("s2s,penalty", [(False, True), (True, False), (True, True)])
def test_replay_feeding_follows_the_script(s2s, penalty):
    """What enters the replay memory, step by step, against the per-agent rules of src/main.py:269-299 written
    out as the script has them (sequential, one agent at a time) and fed with the same per-step facts."""
    import torch
    from collections import Counter
    ro, agent, tr = _setup(E=3, A=12, depth=60, name="Test_2_Level_0", start_eps=0.4, crash_penalty=penalty,
                           switch2switch=s2s, replay_capacity=4000)
    agent.args.batch_size = 10 ** 9                           # never learn: only the feeding is under test
    rp = tr.replay
    called = []
    rp.add = lambda *a, **k: called.append(1)                # the tensor-by-tensor path must not run

    n = ro.E * ro.A
    curr_obs = [None] * n
    buffer_obs = [None] * n                                   # None plays the role of the script's empty list
    buffer_rew = [0.0] * n
    buffer_done = [False] * n
    total = 0
    for t in range(70):
        head = rp.head
        k = tr.step(train=True)
        torch.cuda.synchronize()
        # what the step wrote into the ring (flm_replay_store_slots)
        stored = []
        for j in range(k):
            sl = (head + j) % rp.capacity
            stored.append((_hash(rp.state[sl]), int(rp.action[sl]), float(rp.reward[sl]), _hash(rp.next_state[sl]),
                           bool(rp.done[sl] != 0)))
        update = [False] * n
        for req, obs, _ in (tr.policy.last or []):
            for k, g in enumerate(req.cpu().numpy()):
                update[g] = True
                curr_obs[g] = obs[k]
        net = ro.net.view(-1).cpu().numpy()
        crash = ro.crash.view(-1).cpu().numpy()
        reward = ro.reward.view(-1).cpu().numpy()
        done = ro.done.view(-1).cpu().numpy().astype(bool)
        want = []
        for g in range(n):
            if penalty and crash[g]:
                h = _hash(curr_obs[g])
                want.append((h, 1, -100.0, h, True))
            if not s2s:
                if update[g] and not buffer_done[g]:
                    nxt = ro.consult_observations(torch.tensor([g], dtype=torch.int32, device=ro.device))[0]
                    want.append((_hash(curr_obs[g]), int(net[g]), float(np.float32(reward[g])), _hash(nxt), bool(done[g])))
            else:
                if update[g] and not buffer_done[g]:
                    if buffer_obs[g] is not None:
                        want.append((_hash(buffer_obs[g]), 1, float(np.float32(buffer_rew[g])), _hash(curr_obs[g]),
                                     bool(done[g])))
                        buffer_obs[g] = None
                        buffer_rew[g] = 0.0
                    if net[g] == 0:
                        h = _hash(curr_obs[g])
                        want.append((h, 1, float(np.float32(reward[g])), h, False))
                    elif net[g] == 1:
                        buffer_obs[g] = curr_obs[g].clone()
                if buffer_obs[g] is not None:
                    buffer_rew[g] += reward[g]
            buffer_done[g] = done[g]
        assert Counter(stored) == Counter(want), t
        total += len(want)
    assert total > 100 and not called
    if s2s:
        assert any(x is not None for x in buffer_obs)


def test_default_feeding_goes_through_the_store_kernel():
    """The default mode (src/main.py:275-278) stores a step's transitions with flm_replay_store and writes the
    next-state observations straight into the ring (flm_observations_req_columns_rows): the ring slots written by
    every step, in order and across the wrap-around, against the script's per-agent rule."""
    import torch
    ro, agent, tr = _setup(E=3, A=12, depth=60, name="Test_2_Level_0", start_eps=0.4, replay_capacity=50)
    agent.args.batch_size = 10 ** 9                           # never learn: only the feeding is under test
    rp = tr.replay
    assert tr.curr_obs is None
    called = []
    rp.add = lambda *a, **k: called.append(1)                # the tensor-by-tensor path must not run
    total, wrapped = 0, 0
    for t in range(70):
        head, size = rp.head, rp.size
        k = tr.step(train=True)
        torch.cuda.synchronize()
        want = []
        for req, obs, _ in (tr.policy.last or []):
            net = ro.net.view(-1)[req.long()].cpu().numpy()
            rew = ro.reward.view(-1)[req.long()].to(torch.float32).cpu().numpy()
            dn = ro.done.view(-1)[req.long()].cpu().numpy() != 0
            nxt = ro.consult_observations(req.contiguous())
            for j in range(req.numel()):
                want.append((_hash(obs[j]), int(net[j]), float(rew[j]), _hash(nxt[j]), bool(dn[j])))
        assert k == len(want), t
        assert rp.head == (head + k) % rp.capacity and rp.size == min(rp.capacity, size + k)
        wrapped += head + k > rp.capacity
        for j, w in enumerate(want):
            sl = (head + j) % rp.capacity
            got = (_hash(rp.state[sl]), int(rp.action[sl]), float(rp.reward[sl]), _hash(rp.next_state[sl]),
                   bool(rp.done[sl] != 0))
            assert got == w, (t, j)
        total += k
    assert total > 100 and wrapped > 2 and not called


def test_store_kernel_keeps_the_last_capacity_transitions_of_an_oversized_step():
    """flm_replay_store through the C ABI with more stored requests than ring slots: slots, scalars and rows are those
    of appending one by one to a deque(maxlen=capacity); requests without a row are skipped."""
    import torch
    from flatland_marl_amd._lib import call, ptr
    dev = "cuda"
    rng = np.random.default_rng(5)
    n_agents, cap, head, row_bytes = 200, 23, 17, 64
    perm = rng.permutation(n_agents)
    ready, sw = np.sort(perm[:40]).astype(np.int32), np.sort(perm[40:101]).astype(np.int32)
    n = len(ready) + len(sw)
    rows = 150
    sel_row = rng.integers(0, rows, n).astype(np.int32)
    sel_row[rng.random(n) < 0.25] = -1
    obs = rng.integers(0, 255, (rows, row_bytes)).astype(np.uint8)
    net = rng.integers(0, 2, n_agents).astype(np.int32)
    reward = -rng.random(n_agents)
    done = (rng.random(n_agents) < 0.2).astype(np.int32)
    d = {k: torch.as_tensor(v, device=dev) for k, v in dict(ready=ready, sw=sw, sel_row=sel_row, obs=obs, net=net,
                                                             reward=reward, done=done).items()}
    ring_state = torch.full((cap, row_bytes), 7, dtype=torch.uint8, device=dev)
    ring_action = torch.full((cap,), -5, dtype=torch.int64, device=dev)
    ring_reward = torch.full((cap,), 9.0, dtype=torch.float32, device=dev)
    ring_done = torch.full((cap,), 9.0, dtype=torch.float32, device=dev)
    slot = torch.empty((n,), dtype=torch.int32, device=dev)
    req_out = torch.empty((n,), dtype=torch.int32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    call("flm_replay_store", ptr(d["ready"]), len(ready), ptr(d["sw"]), len(sw), ptr(d["sel_row"]), ptr(d["obs"]), row_bytes,
         ptr(d["net"]), ptr(d["reward"]), ptr(d["done"]), head, cap, ptr(ring_state), ptr(ring_action), ptr(ring_reward),
         ptr(ring_done), ptr(slot), ptr(req_out), ptr(count), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    req = np.concatenate([ready, sw])
    assert np.array_equal(req_out.cpu().numpy(), req)
    ok = np.nonzero(sel_row >= 0)[0]
    assert int(count.item()) == len(ok) > cap
    from collections import deque
    ring = deque(maxlen=cap)
    for i in ok:
        ring.append(i)
    want_slot = np.full(n, -1, np.int32)
    for j, i in enumerate(ring):
        want_slot[i] = (head + j) % cap
    assert np.array_equal(slot.cpu().numpy(), want_slot)
    st, ac, rw, dn = (x.cpu().numpy() for x in (ring_state, ring_action, ring_reward, ring_done))
    for i in ring:
        sl, g = want_slot[i], req[i]
        assert np.array_equal(st[sl], obs[sel_row[i]]) and ac[sl] == net[g] and rw[sl] == np.float32(reward[g])
        assert dn[sl] == float(done[g] != 0)


@pytest.mark.parametrize("s2s,penalty", [(False, True), (True, False), (True, True)])
def test_store_slots_kernel_against_a_sequential_model(s2s, penalty):
    """flm_replay_store_slots through the C ABI over four steps that share the per-agent slots, with a ring smaller than
    one step's transitions: ring rows, scalars, slots and the per-agent state against the rules written out one
    transition at a time (src/main.py:269-299) over a deque(maxlen=capacity)."""
    import torch
    from collections import deque
    from flatland_marl_amd._lib import call, ptr, query
    dev = "cuda"
    rng = np.random.default_rng(11 + 2 * s2s + penalty)
    n_agents, cap, row_bytes, rows = 300, 17, 48, 220
    oversized = 0
    z8 = lambda *sh: np.zeros(sh, np.uint8)
    curr, has_curr, buf, has_buf, buf_rew = z8(n_agents, row_bytes), z8(n_agents), z8(n_agents, row_bytes), z8(n_agents), np.zeros(n_agents)
    ring = deque(maxlen=cap)
    d_curr, d_has_curr, d_buf, d_has_buf = (torch.as_tensor(x, device=dev) for x in (curr, has_curr, buf, has_buf))
    d_buf_rew = torch.zeros(n_agents, dtype=torch.float64, device=dev)
    ring_state = torch.zeros((cap, row_bytes), dtype=torch.uint8, device=dev)
    ring_next = torch.zeros((cap, row_bytes), dtype=torch.uint8, device=dev)
    ring_action = torch.zeros((cap,), dtype=torch.int64, device=dev)
    ring_reward = torch.zeros((cap,), dtype=torch.float32, device=dev)
    ring_done = torch.zeros((cap,), dtype=torch.float32, device=dev)
    head = 5
    slots_of = {}
    for step in range(4):
        perm = rng.permutation(n_agents)
        n1, n2 = (0, 0) if step == 2 else (30, 50)               # one step in which nobody consulted the network
        ready, sw = np.sort(perm[:n1]).astype(np.int32), np.sort(perm[n1:n1 + n2]).astype(np.int32)
        n = n1 + n2
        req = np.concatenate([ready, sw])
        sel_row = rng.integers(0, rows, n).astype(np.int32)
        sel_row[rng.random(n) < 0.2] = -1
        obs = rng.integers(0, 255, (rows, row_bytes)).astype(np.uint8)
        net = rng.integers(0, 2, n_agents).astype(np.int32)
        reward = -rng.random(n_agents)
        done = (rng.random(n_agents) < 0.2).astype(np.int32)
        crash = (rng.random(n_agents) < 0.3).astype(np.int32)
        # the model
        ok = [i for i in range(n) if sel_row[i] >= 0]
        for i in ok:
            curr[req[i]] = obs[sel_row[i]]
            has_curr[req[i]] = 1
        trans = []
        if penalty:
            for g in range(n_agents):
                if crash[g] and has_curr[g]:
                    trans.append((curr[g].copy(), 1, np.float32(-100), curr[g].copy(), 1.0, None))
        if not s2s:
            for i in ok:
                g = req[i]
                trans.append((obs[sel_row[i]].copy(), int(net[g]), np.float32(reward[g]), None, float(done[g] != 0), i))
        else:
            for i in ok:
                g = req[i]
                if has_buf[g]:
                    trans.append((buf[g].copy(), 1, np.float32(buf_rew[g]), curr[g].copy(), float(done[g] != 0), None))
                    has_buf[g] = 0
                    buf_rew[g] = 0.0
            for i in ok:
                g = req[i]
                if net[g] == 0:
                    trans.append((curr[g].copy(), 1, np.float32(reward[g]), curr[g].copy(), 0.0, None))
            for i in ok:
                g = req[i]
                if net[g] == 1:
                    buf[g] = curr[g]
                    has_buf[g] = 1
            buf_rew += np.where(has_buf != 0, reward, 0.0)
        # the kernel
        t = lambda x: torch.as_tensor(x, device=dev)
        d = dict(ready=t(ready), sw=t(sw), sel_row=t(sel_row), obs=t(obs), net=t(net), reward=t(reward), done=t(done), crash=t(crash))
        work = torch.empty((query("flm_replay_store_slots_workspace", n_agents, n),), dtype=torch.int32, device=dev)
        slot = torch.full((max(n, 1),), -7, dtype=torch.int32, device=dev)
        req_out = torch.full((max(n, 1),), -7, dtype=torch.int32, device=dev)
        count = torch.zeros((1,), dtype=torch.int32, device=dev)
        call("flm_replay_store_slots", ptr(d["ready"]) if n1 else None, n1, ptr(d["sw"]) if n2 else None, n2,
             ptr(d["sel_row"]) if n else None, ptr(d["obs"]) if n else None, row_bytes, ptr(d["net"]), ptr(d["reward"]),
             ptr(d["done"]), ptr(d["crash"]), n_agents, int(penalty), int(s2s), ptr(d_curr), ptr(d_has_curr),
             ptr(d_buf) if s2s else None, ptr(d_has_buf) if s2s else None, ptr(d_buf_rew) if s2s else None, head, cap,
             ptr(ring_state), ptr(ring_next), ptr(ring_action), ptr(ring_reward), ptr(ring_done), ptr(work), ptr(slot),
             ptr(req_out), ptr(count), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        k = int(count.item())
        assert k == len(trans), step
        oversized += k > cap                                     # only the last `cap` transitions of such a step survive
        # ring slots of the transitions kept, in order
        kept = trans[-cap:] if k > cap else trans
        first_slot = head
        st, nx = ring_state.cpu().numpy(), ring_next.cpu().numpy()
        ac, rw, dn = ring_action.cpu().numpy(), ring_reward.cpu().numpy(), ring_done.cpu().numpy()
        sl_req = slot.cpu().numpy()
        if n and not s2s:
            assert np.array_equal(req_out.cpu().numpy()[:n], req)
            want_slot = np.full(n, -1, np.int32)
        for j, (s0, a0, r0, s1, d0, i_req) in enumerate(kept):
            sl = (first_slot + j) % cap
            assert np.array_equal(st[sl], s0) and ac[sl] == a0 and rw[sl] == r0 and dn[sl] == d0, (step, j)
            if s1 is not None:
                assert np.array_equal(nx[sl], s1), (step, j)
            else:
                want_slot[i_req] = sl                            # next state left to the observation kernel
        if n and not s2s:
            assert np.array_equal(sl_req[:n], want_slot), step
        head = (head + min(k, cap)) % cap
        assert np.array_equal(d_curr.cpu().numpy(), curr) and np.array_equal(d_has_curr.cpu().numpy(), has_curr), step
        if s2s:
            assert np.array_equal(d_buf.cpu().numpy(), buf) and np.array_equal(d_has_buf.cpu().numpy(), has_buf), step
            assert np.array_equal(d_buf_rew.cpu().numpy(), buf_rew), step
    assert oversized >= 2


def test_policy_select_kernel():
    """flm_policy_select against numpy: greedy choice (eps = 0) is the first maximum over (alternative, action),
    exploring choices (eps = 1) are valid and spread out, sel_row / sel_act name the row acted on."""
    import torch
    from flatland_marl_amd._lib import call, ptr
    rng = np.random.default_rng(3)
    E, A = 6, 50
    n = E * A
    perm = rng.permutation(n)
    ready = np.sort(perm[:70]).astype(np.int32)
    sw = np.sort(perm[70:200]).astype(np.int32)
    cnt = rng.integers(0, 5, len(sw)).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    rows = len(ready) + int(off[-1])
    q = rng.standard_normal((rows, 2)).astype(np.float32)
    q[5] = q[5, 0]                                             # a tie goes to action 0
    dev = "cuda"
    dq, dready, dsw, doff, dcnt = (torch.as_tensor(x, device=dev) for x in (q, ready, sw, off, cnt))
    out = {}
    for eps in (0.0, 1.0):
        net = torch.full((n,), -7, dtype=torch.int32, device=dev)
        best = torch.full((n,), -7, dtype=torch.int32, device=dev)
        sel_row = torch.empty(len(ready) + len(sw), dtype=torch.int32, device=dev)
        sel_act = torch.empty_like(sel_row)
        call("flm_policy_select", ptr(dq), ptr(dready), len(ready), ptr(dsw), len(sw), ptr(doff), ptr(dcnt),
             eps, 1234, 17, ptr(net), ptr(best), ptr(sel_row), ptr(sel_act), torch.cuda.current_stream().cuda_stream)
        out[eps] = [x.cpu().numpy() for x in (net, best, sel_row, sel_act)]
    net, best, sel_row, sel_act = out[0.0]
    untouched = np.setdiff1d(np.arange(n), np.concatenate([ready, sw]))
    assert (net[untouched] == -7).all() and (best[untouched] == -7).all()
    for r, g in enumerate(ready):
        assert net[g] == int(q[r, 1] > q[r, 0]) and sel_row[r] == r and sel_act[r] == net[g]
    for i, g in enumerate(sw):
        k = len(ready) + i
        if cnt[i] == 0:
            assert net[g] == 0 and best[g] == 0 and sel_row[k] == -1
            continue
        base = len(ready) + off[i]
        arg = int(np.argmax(q[base:base + cnt[i]].reshape(-1)))
        assert (net[g], best[g]) == (arg % 2, arg // 2)
        assert sel_row[k] == base + best[g] and sel_act[k] == net[g]
    net, best, sel_row, sel_act = out[1.0]
    has = cnt > 0
    assert set(np.unique(net[ready])) == {0, 1} and set(np.unique(net[sw[has]])) == {0, 1}
    assert (best[sw] >= 0).all() and (best[sw][has] < cnt[has]).all() and (best[sw][~has] == 0).all()
    assert len(np.unique(best[sw[cnt == 4]])) == 4             # every alternative gets drawn
    assert 0.3 < net[ready].mean() < 0.7


def test_replay_sampling_kernel():
    """flm_replay_sample: distinct entries, the same for the same (seed, draw), every entry about equally often;
    flm_replay_gather: the rows those entries name."""
    import torch
    from flatland_marl_amd.replay import DeviceReplay
    dev = "cuda"
    rp = DeviceReplay(5000, (3, 48), dev, seed=11)
    g = torch.Generator().manual_seed(0)
    n = 3210
    state = torch.randint(0, 255, (n, 3, 48), generator=g).to(torch.uint8)
    nxt = torch.randint(0, 255, (n, 3, 48), generator=g).to(torch.uint8)
    act = torch.randint(0, 2, (n,), generator=g)
    rew = torch.randn(n, generator=g)
    done = (torch.rand(n, generator=g) < 0.3).float()
    rp.add(state.cuda(), act.cuda(), rew.cuda(), nxt.cuda(), done.cuda())
    assert len(rp) == n
    counts = np.zeros(n, np.int64)
    B, draws = 512, 300
    seen = set()
    for k in range(draws):
        idx = rp.sample_indices(B).cpu().numpy()
        assert len(np.unique(idx)) == B and idx.min() >= 0 and idx.max() < n            # without replacement
        counts[idx] += 1
        seen.add(idx.tobytes())
    assert len(seen) == draws                                                            # every draw differs
    # every entry is drawn with probability B/n: binomial(draws, B/n), far tails excluded
    p = B / n
    mean, sd = draws * p, (draws * p * (1 - p)) ** 0.5
    assert abs(counts.mean() - mean) < 1e-9 and counts.min() > mean - 6 * sd and counts.max() < mean + 6 * sd
    assert abs(counts[: n // 2].mean() - counts[n // 2:].mean()) < 0.5                   # no drift along the memory
    # the same seed and draw number give the same sample; another seed another one
    a = DeviceReplay(5000, (3, 48), dev, seed=11)
    b = DeviceReplay(5000, (3, 48), dev, seed=12)
    for r_ in (a, b):
        r_.add(state.cuda(), act.cuda(), rew.cuda(), nxt.cuda(), done.cuda())
    rp.draws = 0
    i0, ia, ib = rp.sample_indices(B), a.sample_indices(B), b.sample_indices(B)
    assert torch.equal(i0, ia) and not torch.equal(i0, ib)
    # gather
    rp.draws = 7
    idx = rp.sample_indices(B).cpu().long()
    rp.draws = 7
    s, ac, rw, sn, dn = rp.sample(B)
    assert torch.equal(s.cpu(), state[idx]) and torch.equal(sn.cpu(), nxt[idx]) and torch.equal(ac.cpu(), act[idx])
    assert torch.equal(rw.cpu(), rew[idx]) and torch.equal(dn.cpu(), done[idx])
    # plane-wise gather: the first bytes of every plane arrive, what lies behind them in the buffers is left alone
    for t in rp._out[:1] + rp._out[3:4]:
        t.fill_(9)
    rp.draws = 7
    s2, ac2, rw2, sn2, dn2 = rp.sample(B, used_bytes=20)         # rounded up to 32 of the 48 bytes of a plane
    assert torch.equal(s2[:, :, :32].cpu(), state[idx][:, :, :32]) and torch.equal(sn2[:, :, :32].cpu(), nxt[idx][:, :, :32])
    assert bool((s2[:, :, 32:] == 9).all()) and bool((sn2[:, :, 32:] == 9).all())
    assert torch.equal(ac2.cpu(), act[idx]) and torch.equal(rw2.cpu(), rew[idx]) and torch.equal(dn2.cpu(), done[idx])
    assert torch.equal(rp._out[5].cpu().long(), idx)            # the one-launch form leaves the same list of entries
    # the two-launch forms (flm_replay_sample, then flm_replay_gather / flm_replay_gather_planes): the same rows
    from flatland_marl_amd._lib import call, ptr
    di = idx.to(torch.int32).cuda()
    o = [torch.full_like(rp._out[0], 9), torch.empty_like(ac), torch.empty_like(rw), torch.full_like(rp._out[3], 9), torch.empty_like(dn)]
    call("flm_replay_gather", ptr(rp.state), ptr(rp.next_state), ptr(rp.action), ptr(rp.reward), ptr(rp.done), ptr(di), B,
         rp.row_bytes, ptr(o[0]), ptr(o[3]), ptr(o[1]), ptr(o[2]), ptr(o[4]), 0)
    torch.cuda.synchronize()
    assert torch.equal(o[0].cpu(), state[idx]) and torch.equal(o[3].cpu(), nxt[idx]) and torch.equal(o[1], ac2)
    assert torch.equal(o[2], rw2) and torch.equal(o[4], dn2)
    o[0].fill_(9), o[3].fill_(9)
    call("flm_replay_gather_planes", ptr(rp.state), ptr(rp.next_state), ptr(rp.action), ptr(rp.reward), ptr(rp.done),
         ptr(di), B, rp.row_bytes, 3, 48, 32, ptr(o[0]), ptr(o[3]), ptr(o[1]), ptr(o[2]), ptr(o[4]), 0)
    torch.cuda.synchronize()
    assert torch.equal(o[0], s2) and torch.equal(o[3], sn2) and torch.equal(o[1], ac2)
    # the one-launch form refuses what the two-launch forms refuse
    from flatland_marl_amd._lib import FlmError
    base = (ptr(rp.state), ptr(rp.next_state), ptr(rp.action), ptr(rp.reward), ptr(rp.done))
    outs = (ptr(o[0]), ptr(o[3]), ptr(o[1]), ptr(o[2]), ptr(o[4]), ptr(di))
    for bad in ((n, 11, 7, n + 1, rp.row_bytes, 1, rp.row_bytes, rp.row_bytes),      # a batch larger than the memory
                (n, 11, 7, B, rp.row_bytes, 3, 48, 40),                              # used bytes not a multiple of 16
                (n, 11, 7, B, rp.row_bytes, 4, 48, 32),                              # planes beyond the row
                (0, 11, 7, B, rp.row_bytes, 3, 48, 32)):                             # an empty memory
        with pytest.raises(FlmError):
            call("flm_replay_draw_gather", *base, *bad, *outs, 0)
    # the whole memory can be asked for; more than it holds cannot
    assert sorted(rp.sample_indices(n).cpu().tolist()) == list(range(n))
    with pytest.raises(ValueError):
        rp.sample_indices(n + 1)
    # ring: the oldest entries are overwritten
    rp.add(state[:2000].cuda(), act[:2000].cuda(), rew[:2000].cuda(), nxt[:2000].cuda(), done[:2000].cuda())
    assert len(rp) == 5000 and rp.head == (n + 2000) % 5000
    assert torch.equal(rp.state[n:5000].cpu(), state[:5000 - n]) and torch.equal(rp.state[:rp.head].cpu(), state[5000 - n:2000])


def test_drop_in_replay_buffer_keeps_the_reference_interface():
    """ReplayBuffer(action_space, buffer_size, batch_size, width, height): add / sample / len as src/agent.py:134-198,
    tensors of the reference's shapes and types, sampled transitions are stored ones."""
    import torch
    from flatland_marl_amd.agent import ReplayBuffer
    H, W = 9, 7                                               # 63 values per state: rows are padded to 16 bytes inside
    buf = ReplayBuffer(2, 50, 8, W, H)
    assert len(buf) == 0
    rng = np.random.default_rng(0)
    items = []
    for k in range(70):
        s, sn = rng.integers(-1, 5, (H, W)).astype(np.float64), rng.integers(-1, 5, (H, W)).astype(np.float64)
        items.append((s, k % 2, -0.25 * (k % 4), sn, k % 5 == 0))
        buf.add(*items[-1])
    assert len(buf) == 50                                     # the 20 oldest are gone
    states, actions, rewards, next_states, dones = buf.sample()
    assert states.shape == (8, 1, H, W) and next_states.shape == (8, 1, H, W) and states.dtype == torch.float32
    assert actions.dtype == torch.int64 and rewards.dtype == torch.float32 and dones.dtype == torch.float32
    assert states.is_cuda
    kept = {(x[0].astype(np.float32).tobytes(), x[1], np.float32(x[2]).item(), x[3].astype(np.float32).tobytes(), float(x[4]))
            for x in items[20:]}
    got = set()
    for i in range(8):
        t = (states[i, 0].cpu().numpy().tobytes(), int(actions[i]), float(rewards[i]), next_states[i, 0].cpu().numpy().tobytes(),
             float(dones[i]))
        assert t in kept
        got.add(t)
    assert len(got) == 8
