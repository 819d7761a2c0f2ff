"""Dueling Double-DQN kernels: against the reference's goldens and a plain PyTorch fp32 checker (needs an MI355X).

Tolerance (stated by the task for this floating-point path): Q-values within 1e-5 relative in fp32.  close_q
applies it element by element, |got - want| <= 1e-5 * |want|, to every value above a tenth of the largest |Q| of
the batch; values below that are sums that nearly cancel (Q = value + advantage - mean advantage) and are held
to 1e-5 of that tenth.  The floor is where the reference's own fp32 forward pass still meets the bound against
its float64 result (golden act_q32 vs act_q64: 9e-5 relative on a Q of 5e-4 beside 8e-2, asserted below).
Gradients and sums use close(), relative to the tensor's scale."""
import argparse
import copy
import os

import numpy as np
import pytest
import torch

from tests.golden_util import GOLDEN

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def close(got, want, scale=None, rtol=RTOL):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = np.abs(want).max() if scale is None else scale
    return np.all(np.abs(got - want) <= rtol * np.maximum(np.abs(want), scale))


def close_q(got, want, rtol=RTOL):
    """Per-element relative agreement of Q-values with the float64 golden."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    floor = 0.1 * np.abs(want).max()
    return bool(np.all(np.abs(got - want) <= rtol * np.maximum(np.abs(want), floor)))


def _no_warning():
    import contextlib
    return contextlib.nullcontext()


def tensor_sums(flat, shapes):
    out, off = [], 0
    for shp in shapes:
        n = int(np.prod(shp))
        t = flat[off:off + n].double()
        out.append([float(t.sum()), float(t.abs().sum())])
        off += n
    return np.array(out)


def test_forward_matches_reference_golden():
    from tests import dqn_inputs
    from flatland_marl_amd.model import Dueling_DQN
    gold = dict(np.load(os.path.join(GOLDEN, "dqn_reference.npz")))
    H3, T1 = 3 * dqn_inputs.MAX_RAILS, dqn_inputs.DEPTH + 1
    torch.manual_seed(1234)
    net = Dueling_DQN(T1, H3, 2)
    sums = np.array([[float(p.detach().double().sum()), float(p.detach().double().abs().sum())]
                     for p in net.parameters()])
    # same default init as the reference (float64 sums; only the summation order may differ between hosts)
    assert np.allclose(sums, gold["act_param_sums"], rtol=1e-11, atol=1e-11)
    assert [tuple(v.shape) for v in net.state_dict().values()] == [
        (64, 1, 1, T1), (64,), (512, 64 * H3), (512,), (512, 64 * H3), (512,), (2, 512), (2,), (1, 512), (1,)]
    assert list(net.state_dict()) == ["conv1.weight", "conv1.bias", "fc1_adv.weight", "fc1_adv.bias", "fc1_val.weight",
                                      "fc1_val.bias", "fc2_adv.weight", "fc2_adv.bias", "fc2_val.weight", "fc2_val.bias"]
    net = net.to("cuda")
    x = torch.from_numpy(dqn_inputs.observations(6))[:, None]
    q = net(x).cpu().numpy()
    assert close_q(q, gold["act_q64"])
    assert close_q(gold["act_q32"], gold["act_q64"])        # the reference's own fp32 result sits inside the same band
    # one observation at a time (the act path: batch 1)
    for i in range(6):
        qi = net(x[i:i + 1]).cpu().numpy()
        batch = q.copy()
        batch[i] = qi[0]                                      # judged inside its batch: the scale is the batch's
        assert close_q(batch, gold["act_q64"])


def test_trained_weights_match_reference_golden():
    """Q-values after training has moved the weights: the reference's own agent ran 30 learn steps (lr 1e-3), its
    weights (rounded to float16 so that both sides load the same bits) went back into the reference's network, whose
    float64 Q-values for 64 observations are the golden (oracle/gen_golden_dqn.py trained).  Per element."""
    from tests import dqn_inputs as di
    from flatland_marl_amd.model import Dueling_DQN
    gold = dict(np.load(os.path.join(GOLDEN, "dqn_reference_trained.npz")))
    assert gold["losses"][-1] < 0.5 * gold["losses"][0] and gold["moved_max"].max() > 0.02      # training did move them
    H3, T1 = 3 * di.TRAINED_RAILS, di.DEPTH + 1
    net = Dueling_DQN(T1, H3, 2)
    with torch.no_grad():
        for i, p in enumerate(net.parameters()):
            p.copy_(torch.from_numpy(gold["w%d" % i].view(np.float16).astype(np.float32)).view_as(p))
    net = net.to("cuda")
    obs = di.trained_observations(64)
    x = torch.from_numpy(obs)[:, None]
    q = net(x).cpu().numpy()
    assert close_q(q, gold["q64"])
    assert close_q(gold["q32"], gold["q64"])                  # the reference's own fp32 result, same band
    for i in (0, 5, 63):                                      # the act path, one observation at a time
        batch = q.copy()
        batch[i] = net(x[i:i + 1]).cpu().numpy()[0]
        assert close_q(batch, gold["q64"])
    # the same observations as bytes (what the rollout and the replay memory hold)
    Rp = di.TRAINED_RAILS
    ps = (Rp * T1 + 255) // 256 * 256
    u8 = np.zeros((64, 3, ps), np.uint8)
    u8[:, :, :Rp * T1] = obs.reshape(64, 3, Rp * T1).astype(np.int8).view(np.uint8)
    q8 = net.forward_obs(torch.from_numpy(u8).cuda(), ps).cpu().numpy()
    assert close_q(q8, gold["q64"])


def _args(B):
    from tests import dqn_inputs
    return argparse.Namespace(prediction_depth=dqn_inputs.DEPTH, lr=0.00005, buffer_size=100, batch_size=B,
                              update_every=10, gamma=0.99, tau=1e-3)


def test_learn_matches_reference_golden():
    from tests import dqn_inputs
    from flatland_marl_amd.agent import DQNAgent
    gold = dict(np.load(os.path.join(GOLDEN, "dqn_reference.npz")))
    H3 = 3 * dqn_inputs.MAX_RAILS
    B = 16
    torch.manual_seed(4321)
    agent = DQNAgent(_args(B), action_space=2, bitmap_height=H3)
    s, a, r, sn, d = dqn_inputs.learn_batch(B)
    st, snt = torch.from_numpy(s)[:, None], torch.from_numpy(sn)[:, None]
    assert close_q(agent.qnetwork_local(st).cpu().numpy(), gold["learn_q_before"])
    exp = (st, torch.from_numpy(a), torch.from_numpy(r), snt, torch.from_numpy(d.astype(np.uint8)).float())
    agent.learn(exp, 0.99)
    shapes = [tuple(p.shape) for p in agent.qnetwork_local.parameters()]
    gs = tensor_sums(agent.optimizer.grads, shapes)
    assert close(gs[:, 1], gold["learn_grad_sums"][:, 1], rtol=1e-5)          # sum of |grad| per tensor
    assert close(gs[:, 0], gold["learn_grad_sums"][:, 0], scale=np.abs(gold["learn_grad_sums"][:, 1]).max() * 1e-2,
                 rtol=1e-4)
    assert close(agent.qnetwork_local(st).cpu().numpy(), gold["learn_q_local_after"])
    assert close(agent.qnetwork_target(st).cpu().numpy(), gold["learn_q_target_after"])
    assert close(tensor_sums(agent.qnetwork_local.flat_params(), shapes)[:, 1], gold["learn_local_sums"][:, 1], rtol=1e-6)
    assert close(tensor_sums(agent.qnetwork_target.flat_params(), shapes)[:, 1], gold["learn_target_sums"][:, 1], rtol=1e-6)
    agent.learn(exp, 0.99)
    assert close(agent.qnetwork_local(st).cpu().numpy(), gold["learn2_q_local_after"])
    assert close(tensor_sums(agent.qnetwork_local.flat_params(), shapes)[:, 1], gold["learn2_local_sums"][:, 1], rtol=1e-6)


@pytest.mark.parametrize("B,H3,T1", [(3, 30, 21), (40, 30, 21), (130, 9, 151), (64, 60, 37)])
def test_gradients_elementwise_vs_torch(B, H3, T1):
    """Every gradient element, the loss, the Adam step and the soft update against plain PyTorch fp32."""
    from flatland_marl_amd.agent import DQNAgent
    from tests.torch_ref_dqn import TorchDuelingDQN, learn_step
    torch.manual_seed(7 + B)
    args = argparse.Namespace(prediction_depth=T1 - 1, lr=0.001, buffer_size=10, batch_size=B, update_every=1,
                              gamma=0.9, tau=0.05)
    agent = DQNAgent(args, action_space=2, bitmap_height=H3)
    ref_local = TorchDuelingDQN(T1, H3)
    ref_local.load_state_dict(agent.qnetwork_local.state_dict())
    ref_target = copy.deepcopy(ref_local)
    # make the target differ from the local network
    with torch.no_grad():
        for p in ref_target.parameters():
            p.mul_(0.9)
    agent.qnetwork_target.load_state_dict(ref_target.state_dict())
    opt = torch.optim.Adam(ref_local.parameters(), lr=args.lr)
    g = torch.Generator().manual_seed(B)
    s = torch.randint(-1, 4, (B, 1, H3, T1), generator=g).float()
    sn = torch.randint(-1, 4, (B, 1, H3, T1), generator=g).float()
    a = torch.randint(0, 2, (B,), generator=g)
    r = torch.randn(B, generator=g)
    d = (torch.rand(B, generator=g) < 0.3).float()
    for it in range(2):
        q_ref = ref_local(s).detach().numpy()
        assert close(agent.qnetwork_local(s).cpu().numpy(), q_ref)
        loss = learn_step(ref_local, ref_target, opt, (s, a, r, sn, d), args.gamma, args.tau)
        agent.learn((s, a, r, sn, d), args.gamma)
        assert close(agent.last_loss.cpu().numpy(), loss.detach().numpy())
        off = 0
        flat_g = agent.optimizer.grads.cpu().numpy()
        for p in ref_local.parameters():
            n = p.numel()
            got, want = flat_g[off:off + n], p.grad.reshape(-1).numpy()
            assert close(got, want, rtol=2e-5), (it, off)
            off += n
        # Adam divides by |g|+eps (eps = 1e-8): where a gradient is a near-cancelled sum, its fp32 rounding noise
        # (of the order of 1e-8 absolute in either implementation) decides the size of the step.  Parameters
        # are therefore compared tightly where the gradient is well above that noise, and to within one full
        # step (2*lr, a sign flip) elsewhere.
        g_ref = np.concatenate([p.grad.reshape(-1).numpy() for p in ref_local.parameters()])
        solid = np.abs(g_ref) > 1e-4 * np.abs(g_ref).max()
        got = agent.qnetwork_local.flat_params().cpu().numpy()
        want = np.concatenate([p.detach().reshape(-1).numpy() for p in ref_local.parameters()])
        assert np.all(np.abs(got - want)[solid] <= 1e-3 * args.lr + 1e-6 * np.abs(want[solid]))
        assert np.all(np.abs(got - want) <= 2.0 * args.lr + 1e-6)
        gt = agent.qnetwork_target.flat_params().cpu().numpy()
        wt = np.concatenate([p.detach().reshape(-1).numpy() for p in ref_target.parameters()])
        assert np.all(np.abs(gt - wt) <= args.tau * 2.0 * args.lr + 1e-6 * np.abs(wt) + 1e-7)
        # restart the next iteration from identical weights and Adam moments, so that it tests the second
        # optimizer step (non-zero moments, step count 2) rather than the divergence of two trajectories
        agent.qnetwork_local.load_state_dict(ref_local.state_dict())
        agent.qnetwork_target.load_state_dict(ref_target.state_dict())
        st = opt.state_dict()["state"]
        agent.optimizer.m.copy_(torch.cat([st[i]["exp_avg"].reshape(-1) for i in range(len(st))]))
        agent.optimizer.v.copy_(torch.cat([st[i]["exp_avg_sq"].reshape(-1) for i in range(len(st))]))


@pytest.mark.parametrize("NA,B,H3,T1,double_dqn", [(1, 20, 12, 21, True), (3, 40, 30, 21, True), (5, 130, 9, 151, True),
                                                   (16, 33, 21, 37, True), (4, 24, 15, 31, False),
                                                   (2, 1024, 9, 21, True), (3, 1100, 9, 21, True)])
def test_other_action_spaces_elementwise_vs_torch(NA, B, H3, T1, double_dqn):
    """action_space other than the training script's 2 (the constructor argument of src/model.py:11 and
    src/agent.py:17): state_dict shapes, Q-values, loss and every gradient element against plain PyTorch fp32, for the
    double-DQN target and the reference's plain branch with its [B,B] broadcast.  The reference's own goldens hold two
    actions only: these sizes are checked against PyTorch alone.  (Batches of 1024 and 1100: the largest that the
    one-launch form of the loss and the head's backward pass takes, and the separate kernels beyond it.)"""
    import torch.nn.functional as F
    from flatland_marl_amd.agent import DQNAgent
    from tests.torch_ref_dqn import TorchDuelingDQN
    torch.manual_seed(100 + NA)
    args = argparse.Namespace(prediction_depth=T1 - 1, lr=0.001, buffer_size=10, batch_size=B, update_every=1,
                              gamma=0.9, tau=0.05)
    agent = DQNAgent(args, action_space=NA, bitmap_height=H3, double_dqn=double_dqn)
    sd = agent.qnetwork_local.state_dict()
    assert tuple(sd["fc2_adv.weight"].shape) == (NA, 512) and tuple(sd["fc2_adv.bias"].shape) == (NA,)
    ref_local = TorchDuelingDQN(T1, H3, NA)
    ref_local.load_state_dict(sd)
    ref_target = copy.deepcopy(ref_local)
    with torch.no_grad():
        for p in ref_target.parameters():
            p.mul_(0.9)
    agent.qnetwork_target.load_state_dict(ref_target.state_dict())
    g = torch.Generator().manual_seed(B + NA)
    s = torch.randint(-1, 4, (B, 1, H3, T1), generator=g).float()
    sn = torch.randint(-1, 4, (B, 1, H3, T1), generator=g).float()
    a = torch.randint(0, NA, (B,), generator=g)
    r = torch.randn(B, generator=g)
    d = (torch.rand(B, generator=g) < 0.3).float()
    for n in (1, 5, B):                          # the act path's kernels and the batched ones
        got = agent.qnetwork_local(s[:n]).cpu().numpy()
        assert got.shape == (n, NA)
        assert close(got, ref_local(s[:n]).detach().numpy())
    q = ref_local(s).gather(1, a.unsqueeze(-1)).view(B)
    if double_dqn:
        best = ref_local(sn).max(1)[1]
        q_next = ref_target(sn).gather(1, best.unsqueeze(-1)).view(B)
    else:
        q_next = ref_target(sn).detach().max(1)[0].unsqueeze(-1)
    with pytest.warns(UserWarning) if not double_dqn else _no_warning():
        loss = F.mse_loss(q, r + args.gamma * q_next * (1 - d))
    ref_local.zero_grad()
    loss.backward()
    agent.learn((s, a, r, sn, d), args.gamma)
    assert close(agent.last_loss.cpu().numpy(), loss.detach().numpy())
    off = 0
    flat_g = agent.optimizer.grads.cpu().numpy()
    for name, p in ref_local.named_parameters():
        n = p.numel()
        assert close(flat_g[off:off + n], p.grad.reshape(-1).numpy(), rtol=2e-5), name
        off += n
    assert off == flat_g.size


def test_action_space_limits():
    from flatland_marl_amd._lib import DQN_MAX_ACTIONS
    from flatland_marl_amd.model import Dueling_DQN
    for bad in (0, DQN_MAX_ACTIONS + 1):
        with pytest.raises(ValueError):
            Dueling_DQN(21, 12, bad)


@pytest.mark.parametrize("H3,T1", [(21, 37), (300, 151), (12, 301)])
def test_small_batch_forward_vs_torch(H3, T1):
    """The act path (batches of 1..8: the four-rows-per-workgroup convolution and the weight-streaming first dense
    layer) against plain PyTorch fp32, for every batch size, with rows that straddle samples (H3 not a multiple of 4)."""
    from flatland_marl_amd.agent import DQNAgent
    from tests.torch_ref_dqn import TorchDuelingDQN
    torch.manual_seed(H3)
    args = argparse.Namespace(prediction_depth=T1 - 1, lr=0.001, buffer_size=10, batch_size=8, update_every=1,
                              gamma=0.9, tau=0.05)
    agent = DQNAgent(args, action_space=2, bitmap_height=H3)
    ref = TorchDuelingDQN(T1, H3)
    ref.load_state_dict(agent.qnetwork_local.state_dict())
    g = torch.Generator().manual_seed(H3 + T1)
    for B in range(1, 10):                      # 9: the first batch size beyond the act path
        x = torch.randint(-1, 6, (B, 1, H3, T1), generator=g).float()
        want = ref(x).detach().numpy()
        got = agent.qnetwork_local(x).cpu().numpy()
        assert got.shape == (B, 2)
        assert close(got, want), B


@pytest.mark.parametrize("B,H3,T1", [(64, 30, 40), (200, 45, 151)])
def test_forward_wide_dynamic_range_vs_float64(B, H3, T1):
    """The matrix products split every fp32 operand into three bf16 terms and leave out the partial products below
    2^-21 of a product.  With inputs and weights spread over sixteen binary orders of magnitude the Q-values must
    stay as close to the float64 result as PyTorch's own fp32 forward pass does (within a factor of four of its
    error, plus 1e-6 of the largest |Q|)."""
    from flatland_marl_amd.model import Dueling_DQN
    from tests.torch_ref_dqn import TorchDuelingDQN
    torch.manual_seed(B + H3)
    net = Dueling_DQN(T1, H3, 2)
    g = torch.Generator().manual_seed(7 * B + T1)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(torch.exp2(torch.randint(-12, 5, p_.shape, generator=g).float()))
    ref32 = TorchDuelingDQN(T1, H3)
    ref32.load_state_dict(net.state_dict())
    ref64 = copy.deepcopy(ref32).double()
    x = torch.randn(B, 1, H3, T1, generator=g) * torch.exp2(torch.randint(-10, 7, (B, 1, H3, T1), generator=g).float())
    q64 = ref64(x.double()).detach().numpy()
    q32 = ref32(x).detach().numpy().astype(np.float64)
    got = net.to("cuda")(x).cpu().numpy().astype(np.float64)
    scale = np.abs(q64).max()
    err32, err = np.abs(q32 - q64).max(), np.abs(got - q64).max()
    assert np.isfinite(got).all()
    assert err <= 4 * err32 + 1e-6 * scale, (err, err32, scale)


@pytest.mark.parametrize("n,Rp,T1,ps,off", [(7, 5, 13, 80, 0), (3, 70, 151, 10752, 0), (22000, 2, 5, 16, 0), (9, 4, 9, 48, 1)])
def test_byte_planes_to_float_rows(n, Rp, T1, ps, off):
    """flm_obs_to_f32: plane 0 is read as int8, the heat planes as uint8; rows of Rp*T1 values are packed without the
    plane padding.  Covers planes shorter than a workgroup's piece, more than 65,535 planes (two launches), and a
    source that is not word aligned."""
    from flatland_marl_amd._lib import call, ptr
    g = torch.Generator().manual_seed(n + Rp)
    raw = torch.randint(0, 256, (n * 3 * ps + 4,), generator=g, dtype=torch.uint8).cuda()
    obs = raw[off:off + n * 3 * ps]
    out = torch.full((n * 3 * Rp * T1 + 8,), -7.0, device="cuda")
    call("flm_obs_to_f32", ptr(obs), n, Rp, T1, ps, ptr(out), torch.cuda.current_stream().cuda_stream)
    planes = obs.view(n, 3, ps)[:, :, :Rp * T1].cpu().numpy()
    want = planes.astype(np.float32)
    want[:, 0] = planes[:, 0].view(np.int8).astype(np.float32)
    got = out.cpu().numpy()
    assert np.array_equal(got[:n * 3 * Rp * T1].reshape(n, 3, Rp * T1), want)
    assert np.all(got[n * 3 * Rp * T1:] == -7.0)          # nothing behind the last row is written


@pytest.mark.parametrize("B,Rp,T1,pad", [(5, 10, 21, 0), (300, 7, 151, 3), (33, 4, 301, 1), (700, 100, 151, 0),
                                         (90, 6, 165, 2), (41, 9, 201, 1), (37, 5, 251, 0)])
def test_forward_from_uint8_observations_vs_torch(B, Rp, T1, pad):
    """The byte-input forward (convolution reads int8 / uint8 planes) against plain PyTorch fp32 on the same values;
    plane strides that leave the rows at every byte alignment, horizons beyond 256 steps, ragged row tiles."""
    from flatland_marl_amd.agent import DQNAgent
    from tests.torch_ref_dqn import TorchDuelingDQN
    torch.manual_seed(B)
    H3 = 3 * Rp
    args = argparse.Namespace(prediction_depth=T1 - 1, lr=0.001, buffer_size=10, batch_size=B, update_every=1,
                              gamma=0.9, tau=0.05)
    agent = DQNAgent(args, action_space=2, bitmap_height=H3)
    ref = TorchDuelingDQN(T1, H3)
    ref.load_state_dict(agent.qnetwork_local.state_dict())
    ps = (Rp * T1 + 15) // 16 * 16 + 16 * pad
    g = torch.Generator().manual_seed(B + 1)
    obs = torch.zeros((B, 3, ps), dtype=torch.uint8)
    own = torch.randint(-1, 2, (B, Rp * T1), generator=g)
    obs[:, 0, :Rp * T1] = own.to(torch.int8).view(torch.uint8)
    obs[:, 1, :Rp * T1] = torch.randint(0, 200, (B, Rp * T1), generator=g).to(torch.uint8)
    obs[:, 2, :Rp * T1] = torch.randint(0, 255, (B, Rp * T1), generator=g).to(torch.uint8)
    obs[:, :, Rp * T1:] = 77                     # bytes between planes must not leak into the result
    x = torch.cat([own.float(), obs[:, 1, :Rp * T1].float(), obs[:, 2, :Rp * T1].float()], dim=1).view(B, 1, H3, T1)
    want = ref(x).detach().numpy()
    got = agent.qnetwork_local.forward_obs(obs.cuda(), ps).cpu().numpy()
    assert close(got, want)


@pytest.mark.parametrize("Rp,T1,B,ps", [(37, 61, 96, 2304), (11, 151, 70, 1664 + 16), (5, 255, 33, 1280), (1, 100, 9, 112),
                                        (7, 159, 300, 1120), (6, 31, 40, 192), (9, 32, 261, 288), (5, 160, 20, 800),
                                        (23, 96, 17, 2208)])
def test_learn_from_uint8_observations_matches_float_learn(Rp, T1, B, ps):
    """flm_dqn_learn_grads_u8 (bytes in, as the device replay memory stores them) against flm_dqn_learn_grads on
    the float expansion of the same observations: loss and every gradient within 1e-5 relative, the convolution's
    own gradients also within 1e-5 of their own largest.  The byte form of the convolution's weight gradient takes
    horizons up to 159 columns (one to five column tiles, the bias column first or last in its tile: 31, 32, 96, 159),
    longer ones (160, 255) go through the float copy; batches below and above the 256 workgroups of that kernel;
    plane strides that are and are not multiples of 256; a single rail per plane; heat counts up to 255."""
    import types
    import torch
    from flatland_marl_amd.agent import DQNAgent
    from flatland_marl_amd._lib import call, ptr, query
    assert ps >= Rp * T1 and ps % 16 == 0
    args = types.SimpleNamespace(prediction_depth=T1 - 1, lr=5e-5, buffer_size=10, batch_size=B, update_every=10,
                                 gamma=0.99, tau=1e-3)
    torch.manual_seed(1)
    agent = DQNAgent(args, action_space=2, bitmap_height=3 * Rp)
    g = torch.Generator().manual_seed(2)
    obs = torch.zeros((2, B, 3, ps), dtype=torch.uint8)
    obs[:, :, 0, :Rp * T1] = torch.randint(-1, 2, (2, B, Rp * T1), generator=g).to(torch.int8).view(torch.uint8)
    obs[:, :, 1:, :Rp * T1] = torch.randint(0, 4, (2, B, 2, Rp * T1), generator=g).to(torch.uint8)
    obs[:, :, 2, :Rp * T1:7] = torch.randint(0, 256, obs[:, :, 2, :Rp * T1:7].shape, generator=g).to(torch.uint8)
    obs = obs.cuda()
    f = torch.empty((2, B, 3 * Rp, T1), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for k in range(2):
        call("flm_obs_to_f32", ptr(obs[k]), B, Rp, T1, ps, ptr(f[k]), st)
    act = torch.randint(0, 2, (B,), generator=g).cuda()
    rew = torch.randn(B, generator=g).cuda()
    done = (torch.rand(B, generator=g) < 0.2).float().cuda()
    local, target = agent.qnetwork_local.flat_params(), agent.qnetwork_target.flat_params()
    target.mul_(1.01)                                           # make the two networks differ
    grads = torch.zeros((2, local.numel()), device="cuda")
    loss = torch.zeros((2, 1), device="cuda")
    ws = torch.empty(query("flm_dqn_learn_workspace_u8", B, Rp, T1, 2), device="cuda")
    call("flm_dqn_learn_grads", ptr(local), ptr(target), ptr(f[0]), ptr(f[1]), ptr(act), ptr(rew), ptr(done), 0.99, 1, B,
         3 * Rp, T1, 2, ptr(ws), ptr(grads[0]), ptr(loss[0]), st)
    call("flm_dqn_learn_grads_u8", ptr(local), ptr(target), ptr(obs[0]), ptr(obs[1]), ps, ptr(act), ptr(rew), ptr(done),
         0.99, 1, B, Rp, T1, 2, ptr(ws), ptr(grads[1]), ptr(loss[1]), st)
    torch.cuda.synchronize()
    assert torch.allclose(loss[0], loss[1], rtol=1e-5)
    scale = grads[0].abs().max().item()
    assert scale > 0 and (grads[0] - grads[1]).abs().max().item() <= 1e-5 * scale
    n_conv = query("flm_dqn_grad_split", 3 * Rp, T1, 2)          # the convolution's weights and bias come first
    conv_scale = grads[0, :n_conv].abs().max().item()
    assert conv_scale > 0 and (grads[0, :n_conv] - grads[1, :n_conv]).abs().max().item() <= 1e-5 * conv_scale


def test_learn_shape_of_the_benchmark_elementwise_vs_torch():
    """The learn step at the size the training loop runs it (batch 512, max_rails 100, depth 150): loss, Q-values and
    every gradient element through flm_dqn_learn_grads (float states) and flm_dqn_learn_grads_u8 (the replay
    memory's bytes) against plain PyTorch fp32 on the same values."""
    from flatland_marl_amd.agent import DQNAgent
    from flatland_marl_amd._lib import call, ptr, query
    from tests.torch_ref_dqn import TorchDuelingDQN, learn_step
    B, Rp, T1 = 512, 100, 151
    H3 = 3 * Rp
    ps = (Rp * T1 + 255) // 256 * 256
    args = argparse.Namespace(prediction_depth=T1 - 1, lr=5e-5, buffer_size=10, batch_size=B, update_every=1,
                              gamma=0.99, tau=1e-3)
    torch.manual_seed(99)
    agent = DQNAgent(args, action_space=2, bitmap_height=H3)
    ref_local = TorchDuelingDQN(T1, H3)
    ref_local.load_state_dict(agent.qnetwork_local.state_dict())
    ref_target = copy.deepcopy(ref_local)
    with torch.no_grad():
        for p in ref_target.parameters():
            p.mul_(0.97)
    agent.qnetwork_target.load_state_dict(ref_target.state_dict())
    # observations shaped like real ones: a sparse own plane in {-1,0,1}, heat counts on the same sparse support
    g = torch.Generator().manual_seed(5)
    obs = torch.zeros((2, B, 3, ps), dtype=torch.uint8)
    mask = torch.rand((2, B, Rp * T1), generator=g) < 0.02
    own = torch.where(mask, torch.randint(0, 2, mask.shape, generator=g) * 2 - 1, torch.zeros(mask.shape, dtype=torch.long))
    obs[:, :, 0, :Rp * T1] = own.to(torch.int8).view(torch.uint8)
    heat = torch.rand((2, B, 2, Rp * T1), generator=g) < 0.05
    obs[:, :, 1:, :Rp * T1] = torch.where(heat, torch.randint(1, 9, heat.shape, generator=g), torch.zeros(heat.shape, dtype=torch.long)).to(torch.uint8)
    x = torch.cat([own.float(), obs[:, :, 1, :Rp * T1].float(), obs[:, :, 2, :Rp * T1].float()], dim=2).view(2, B, 1, H3, T1)
    a = torch.randint(0, 2, (B,), generator=g)
    r = -torch.rand(B, generator=g)
    d = (torch.rand(B, generator=g) < 0.1).float()
    opt = torch.optim.Adam(ref_local.parameters(), lr=args.lr)
    q_ref = ref_local(x[0]).detach().numpy()
    loss_ref = learn_step(ref_local, ref_target, opt, (x[0], a, r, x[1], d), args.gamma, args.tau).detach().numpy()
    g_ref = [p.grad.reshape(-1).numpy() for p in ref_local.parameters()]
    assert close_q(agent.qnetwork_local(x[0]).cpu().numpy(), q_ref)
    assert close_q(agent.qnetwork_local.forward_obs(obs[0].cuda(), ps).cpu().numpy(), q_ref)
    local, target = agent.qnetwork_local.flat_params(), agent.qnetwork_target.flat_params()
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(query("flm_dqn_learn_workspace_u8", B, Rp, T1, 2), device="cuda")
    dev = lambda t: t.cuda().contiguous()
    da, dr, dd, dx, dobs = dev(a), dev(r), dev(d), dev(x.view(2, B, H3, T1)), dev(obs)
    for which in ("float", "bytes"):
        grads = torch.zeros_like(local)
        loss = torch.zeros(1, device="cuda")
        if which == "float":
            call("flm_dqn_learn_grads", ptr(local), ptr(target), ptr(dx[0]), ptr(dx[1]), ptr(da), ptr(dr), ptr(dd), args.gamma,
                 1, B, H3, T1, 2, ptr(ws), ptr(grads), ptr(loss), st)
        else:
            call("flm_dqn_learn_grads_u8", ptr(local), ptr(target), ptr(dobs[0]), ptr(dobs[1]), ps, ptr(da), ptr(dr), ptr(dd),
                 args.gamma, 1, B, Rp, T1, 2, ptr(ws), ptr(grads), ptr(loss), st)
        torch.cuda.synchronize()
        assert close(loss.cpu().numpy(), loss_ref), which
        flat, off = grads.cpu().numpy(), 0
        for i, want in enumerate(g_ref):
            got = flat[off:off + want.size]
            assert close(got, want, rtol=2e-5), (which, i)
            off += want.size


def test_plain_dqn_matches_reference_golden():
    """double_dqn=False: the reference's own learn step (src/agent.py:103-105, with its [B,B] broadcast of the targets
    inside mse_loss) recorded by oracle/gen_golden_dqn.py, and the same step in plain PyTorch, element by element."""
    from tests import dqn_inputs
    from flatland_marl_amd.agent import DQNAgent
    from tests.torch_ref_dqn import TorchDuelingDQN
    import torch.nn.functional as F
    gold = dict(np.load(os.path.join(GOLDEN, "dqn_reference_plain.npz")))
    H3 = 3 * dqn_inputs.MAX_RAILS
    B = 16
    torch.manual_seed(4321)
    agent = DQNAgent(_args(B), action_space=2, bitmap_height=H3, double_dqn=False)
    agent.qnetwork_target.flat_params().mul_(1.05)
    ref_local = TorchDuelingDQN(dqn_inputs.DEPTH + 1, H3)
    ref_local.load_state_dict(agent.qnetwork_local.state_dict())
    ref_target = TorchDuelingDQN(dqn_inputs.DEPTH + 1, H3)
    ref_target.load_state_dict(agent.qnetwork_target.state_dict())
    s, a, r, sn, d = dqn_inputs.learn_batch(B)
    st, snt = torch.from_numpy(s)[:, None], torch.from_numpy(sn)[:, None]
    at, rt, dt = torch.from_numpy(a), torch.from_numpy(r), torch.from_numpy(d.astype(np.uint8)).float()
    agent.learn((st, at, rt, snt, dt), 0.99)
    assert close(agent.last_loss.cpu().numpy(), gold["loss"])
    shapes = [tuple(p.shape) for p in agent.qnetwork_local.parameters()]
    gs = tensor_sums(agent.optimizer.grads, shapes)
    assert close(gs[:, 1], gold["grad_sums"][:, 1], rtol=1e-5)
    assert close_q(agent.qnetwork_local(st).cpu().numpy(), gold["q_local_after"])
    assert close_q(agent.qnetwork_target(st).cpu().numpy(), gold["q_target_after"])
    assert close(tensor_sums(agent.qnetwork_local.flat_params(), shapes)[:, 1], gold["local_sums"][:, 1], rtol=1e-6)
    # the same in plain PyTorch, every gradient element
    q = ref_local(st).gather(1, at.unsqueeze(-1)).view(B)
    q_next = ref_target(snt).detach().max(1)[0].unsqueeze(-1)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        loss = F.mse_loss(q, rt + 0.99 * q_next * (1 - dt))
    loss.backward()
    assert close(agent.last_loss.cpu().numpy(), loss.detach().numpy())
    flat, off = agent.optimizer.grads.cpu().numpy(), 0
    for p in ref_local.parameters():
        want = p.grad.reshape(-1).numpy()
        assert close(flat[off:off + want.size], want, rtol=2e-5)
        off += want.size


@pytest.mark.parametrize("u8", [False, True])
def test_learn_step_in_two_phases_gives_the_same_bits(u8):
    """flm_dqn_learn_grads(_u8)_phase 1 then 2 = the one-call form bit for bit, and the dense layers' gradients are
    already final after phase 1 (what a data-parallel caller sends while phase 2 runs)."""
    from flatland_marl_amd._lib import call, ptr, query
    dev = "cuda"
    B, Rp, T1, NA = 70, 12, 37, 2
    H3 = 3 * Rp
    ps = (Rp * T1 + 255) // 256 * 256
    g = torch.Generator().manual_seed(5)
    n = query("flm_dqn_param_count", H3, T1, NA)
    local = (torch.randn(n, generator=g) * 0.05).to(dev)
    target = (torch.randn(n, generator=g) * 0.05).to(dev)
    act = torch.randint(0, NA, (B,), generator=g).to(dev)
    rew = torch.randn(B, generator=g).to(dev)
    done = (torch.rand(B, generator=g) < 0.2).float().to(dev)
    st = torch.cuda.current_stream().cuda_stream
    split = query("flm_dqn_grad_split", H3, T1, NA)
    assert split == 64 * T1 + 64
    if u8:
        s8 = torch.randint(0, 3, (B, 3, ps), generator=g, dtype=torch.uint8).to(dev)
        sn8 = torch.randint(0, 3, (B, 3, ps), generator=g, dtype=torch.uint8).to(dev)
        ws = torch.empty(query("flm_dqn_learn_workspace_u8", B, Rp, T1, NA), device=dev)
        head = (ptr(local), ptr(target), ptr(s8), ptr(sn8), ps, ptr(act), ptr(rew), ptr(done), 0.97, 1, B, Rp, T1, NA, ptr(ws))
        name = "flm_dqn_learn_grads_u8"
    else:
        s = torch.randint(-1, 3, (B, H3, T1), generator=g).float().to(dev)
        sn = torch.randint(-1, 3, (B, H3, T1), generator=g).float().to(dev)
        ws = torch.empty(query("flm_dqn_learn_workspace", B, H3, T1, NA), device=dev)
        head = (ptr(local), ptr(target), ptr(s), ptr(sn), ptr(act), ptr(rew), ptr(done), 0.97, 1, B, H3, T1, NA, ptr(ws))
        name = "flm_dqn_learn_grads"
    want, loss_w = torch.zeros(n, device=dev), torch.zeros(1, device=dev)
    call(name, *head, ptr(want), ptr(loss_w), st)
    got, loss_g = torch.full((n,), 3.0, device=dev), torch.zeros(1, device=dev)
    call(name + "_phase", *head, ptr(got), ptr(loss_g), 1, st)
    torch.cuda.synchronize()
    assert torch.equal(got[split:], want[split:]) and torch.equal(loss_g, loss_w)
    assert bool((got[:split] == 3.0).all())                   # phase 1 leaves the convolution's gradients alone
    call(name + "_phase", *head, ptr(got), ptr(loss_g), 2, st)
    torch.cuda.synchronize()
    assert torch.equal(got, want)


@pytest.mark.parametrize("u8", [True, False])
def test_learn_step_with_next_states_behind_the_states_gives_the_same_bits(u8):
    """When the next states lie right behind the states in memory (the device replay memory's layout) the local
    network's two convolutions are one launch: the same gradients and loss bit for bit as from separate buffers."""
    from flatland_marl_amd._lib import call, ptr, query
    dev = "cuda"
    B, Rp, T1, NA = 45, 11, 53, 2
    H3 = 3 * Rp
    ps = (Rp * T1 + 255) // 256 * 256
    g = torch.Generator().manual_seed(9)
    n = query("flm_dqn_param_count", H3, T1, NA)
    local = (torch.randn(n, generator=g) * 0.05).to(dev)
    target = (torch.randn(n, generator=g) * 0.05).to(dev)
    act = torch.randint(0, NA, (B,), generator=g).to(dev)
    rew = torch.randn(B, generator=g).to(dev)
    done = (torch.rand(B, generator=g) < 0.2).float().to(dev)
    st = torch.cuda.current_stream().cuda_stream
    if u8:
        both = torch.randint(0, 3, (2 * B, 3, ps), generator=g, dtype=torch.uint8).to(dev)
        ws = torch.empty(query("flm_dqn_learn_workspace_u8", B, Rp, T1, NA), device=dev)
        name = "flm_dqn_learn_grads_u8"
        args = lambda s, sn: (ptr(local), ptr(target), ptr(s), ptr(sn), ps, ptr(act), ptr(rew), ptr(done), 0.97, 1, B, Rp, T1,
                              NA, ptr(ws))
    else:
        both = torch.randint(-1, 3, (2 * B, H3, T1), generator=g).float().to(dev)
        ws = torch.empty(query("flm_dqn_learn_workspace", B, H3, T1, NA), device=dev)
        name = "flm_dqn_learn_grads"
        args = lambda s, sn: (ptr(local), ptr(target), ptr(s), ptr(sn), ptr(act), ptr(rew), ptr(done), 0.97, 1, B, H3, T1, NA,
                              ptr(ws))
    apart_s, apart_n = both[:B].clone(), both[B:].clone()
    want, loss_w = torch.zeros(n, device=dev), torch.zeros(1, device=dev)
    call(name, *args(apart_s, apart_n), ptr(want), ptr(loss_w), st)
    got, loss_g = torch.zeros(n, device=dev), torch.zeros(1, device=dev)
    call(name, *args(both[:B], both[B:]), ptr(got), ptr(loss_g), st)
    torch.cuda.synchronize()
    assert both[B:].data_ptr() == both.data_ptr() + B * both[0].numel() * both.element_size()
    assert torch.equal(got, want) and torch.equal(loss_g, loss_w)
    assert float(want.abs().max()) > 0


@pytest.mark.parametrize("B,Rp,T1", [(64, 20, 37), (130, 3, 151), (96, 12, 96)])
def test_gradient_error_against_float64_is_of_pytorch_fp32_size(B, Rp, T1):
    """The gradients of a learn step from float states and from uint8 observations (where the convolution's weight
    gradient comes from the bytes) against the same step in float64: per parameter tensor the largest error stays within
    four times the largest error of plain PyTorch fp32 on the same inputs plus 5e-7 of the tensor's largest gradient
    (four fp32 roundings of it: where PyTorch's own sum happens to come out almost exact, a multiple of its error is no measure).
    The operand split of the matrix kernels therefore loses nothing that an fp32 implementation keeps."""
    from flatland_marl_amd.agent import DQNAgent
    from flatland_marl_amd._lib import call, ptr, query
    from tests.torch_ref_dqn import TorchDuelingDQN
    import torch.nn.functional as F
    H3 = 3 * Rp
    ps = (Rp * T1 + 255) // 256 * 256
    torch.manual_seed(31 + B)
    args = argparse.Namespace(prediction_depth=T1 - 1, lr=0.001, buffer_size=10, batch_size=B, update_every=1,
                              gamma=0.9, tau=0.05)
    agent = DQNAgent(args, action_space=2, bitmap_height=H3)
    # weights spread over a few orders of magnitude, as training leaves them
    with torch.no_grad():
        flat = agent.qnetwork_local.flat_params()
        flat.mul_(torch.exp(2.0 * torch.randn(flat.shape, generator=torch.Generator().manual_seed(1))).to(flat.device))
    ref32 = TorchDuelingDQN(T1, H3)
    ref32.load_state_dict(agent.qnetwork_local.state_dict())
    tgt32 = copy.deepcopy(ref32)
    with torch.no_grad():
        for p in tgt32.parameters():
            p.mul_(0.9)
    agent.qnetwork_target.load_state_dict(tgt32.state_dict())
    ref64, tgt64 = copy.deepcopy(ref32).double(), copy.deepcopy(tgt32).double()
    g = torch.Generator().manual_seed(B)
    obs = torch.zeros((2, B, 3, ps), dtype=torch.uint8)
    obs[:, :, 0, :Rp * T1] = torch.randint(-1, 2, (2, B, Rp * T1), generator=g).to(torch.int8).view(torch.uint8)
    obs[:, :, 1:, :Rp * T1] = torch.randint(0, 12, (2, B, 2, Rp * T1), generator=g).to(torch.uint8)
    x = torch.cat([obs[:, :, 0, :Rp * T1].view(torch.int8).float(), obs[:, :, 1, :Rp * T1].float(),
                   obs[:, :, 2, :Rp * T1].float()], dim=2).view(2, B, 1, H3, T1)
    a = torch.randint(0, 2, (B,), generator=g)
    r = torch.randn(B, generator=g)
    d = (torch.rand(B, generator=g) < 0.3).float()

    def grads_of(local, target, s, sn, rr, dd):
        q = local(s).gather(1, a.unsqueeze(-1)).squeeze(-1)
        best = local(sn).max(1)[1]
        q_next = target(sn).gather(1, best.unsqueeze(-1)).squeeze(-1).detach()
        loss = F.mse_loss(q, rr + 0.9 * q_next * (1 - dd))
        local.zero_grad()
        loss.backward()
        return [p.grad.reshape(-1).double().numpy() for p in local.parameters()]

    g64 = grads_of(ref64, tgt64, x[0].double(), x[1].double(), r.double(), d.double())
    g32 = grads_of(ref32, tgt32, x[0], x[1], r, d)
    # the same action choice in both precisions, or the comparison is between two different losses
    assert torch.equal(ref64(x[1].double()).max(1)[1], ref32(x[1]).max(1)[1])
    dev = "cuda"
    local, target = agent.qnetwork_local.flat_params(), agent.qnetwork_target.flat_params()
    st = torch.cuda.current_stream().cuda_stream
    out = torch.zeros((2, local.numel()), device=dev)
    loss = torch.zeros((2, 1), device=dev)
    ws = torch.empty(query("flm_dqn_learn_workspace_u8", B, Rp, T1, 2), device=dev)
    xs = x.to(dev).view(2, B, H3, T1).contiguous()
    o8 = obs.to(dev)
    ad, rd, dd = a.to(dev), r.to(dev), d.to(dev)
    call("flm_dqn_learn_grads", ptr(local), ptr(target), ptr(xs[0]), ptr(xs[1]), ptr(ad), ptr(rd), ptr(dd), 0.9, 1, B, H3, T1,
         2, ptr(ws), ptr(out[0]), ptr(loss[0]), st)
    call("flm_dqn_learn_grads_u8", ptr(local), ptr(target), ptr(o8[0]), ptr(o8[1]), ps, ptr(ad), ptr(rd), ptr(dd), 0.9, 1, B,
         Rp, T1, 2, ptr(ws), ptr(out[1]), ptr(loss[1]), st)
    torch.cuda.synchronize()
    got = out.cpu().double().numpy()
    worst = 0.0
    for which in range(2):
        off = 0
        for w64, w32 in zip(g64, g32):
            n = w64.size
            e_ours = np.abs(got[which, off:off + n] - w64).max()
            e_t32 = np.abs(w32 - w64).max()
            top = np.abs(w64).max()
            assert e_ours <= 4.0 * e_t32 + 5e-7 * top, (which, off, e_ours, e_t32, top)
            worst = max(worst, e_ours / max(e_t32, 1e-30))
            off += n
    assert worst > 0


@pytest.mark.parametrize("Rp,T1,B,rails", [(70, 151, 300, 17), (100, 151, 64, 14), (70, 151, 2250, 9), (23, 61, 40, 5),
                                           (70, 151, 33, 61), (70, 151, 33, 64), (16, 37, 20, 8), (70, 151, 5, 17)])
def test_forward_over_the_used_rails_matches_the_plain_forward(Rp, T1, B, rails):
    """flm_dqn_forward_u8_rails (rails from `rails` on are empty in every plane: they are not computed but enter through
    the first dense layer's bias) against flm_dqn_forward_u8 on the same observations: Q-values within 1e-5 per element
    (close_q).  Rail counts that are and are not multiples of four, too few empty rails to bother (the plain form runs),
    a small batch (plain form), weights with conv biases of both signs so that relu(bias) is zero for some filters."""
    import types
    from flatland_marl_amd.agent import DQNAgent
    ps = (Rp * T1 + 255) // 256 * 256
    args = types.SimpleNamespace(prediction_depth=T1 - 1, lr=5e-5, buffer_size=10, batch_size=B, update_every=10,
                                 gamma=0.99, tau=1e-3)
    torch.manual_seed(3)
    agent = DQNAgent(args, action_space=2, bitmap_height=3 * Rp)
    net = agent.qnetwork_local
    with torch.no_grad():                                      # weights as training leaves them: spread, biases of both signs
        flat = net.flat_params()
        flat.mul_(torch.exp(torch.randn(flat.shape, generator=torch.Generator().manual_seed(1))).to(flat.device))
        sd = net.state_dict()
        assert (sd["conv1.bias"] > 0).any() and (sd["conv1.bias"] < 0).any()
    g = torch.Generator().manual_seed(B)
    obs = torch.zeros((B, 3, Rp, T1), dtype=torch.uint8)
    obs[:, 0, :rails] = torch.randint(-1, 2, (B, rails, T1), generator=g).to(torch.int8).view(torch.uint8)
    obs[:, 1:, :rails] = torch.randint(0, 30, (B, 2, rails, T1), generator=g).to(torch.uint8)
    buf = torch.full((B, 3, ps), 91, dtype=torch.uint8)       # bytes between the planes must not matter
    buf[:, :, :Rp * T1] = obs.view(B, 3, Rp * T1)
    buf = buf.cuda()
    want = net.forward_obs(buf, ps).cpu().numpy()
    got = net.forward_obs(buf, ps, rails_used=rails).cpu().numpy()
    assert close_q(got, want)
    assert np.abs(want).max() > 0
    # what the empty rows hold is not read: garbage there leaves the result of the rails form unchanged
    buf2 = buf.clone()
    buf2.view(B, 3, ps)[:, :, ((rails + 3) & ~3) * T1:Rp * T1] = 201
    if ((rails + 3) & ~3) + 4 <= Rp and B > 8:
        got2 = net.forward_obs(buf2, ps, rails_used=rails).cpu().numpy()
        assert np.array_equal(got2, got)


@pytest.mark.parametrize("Rp,T1,B,rails,ddqn", [(70, 151, 64, 17, 1), (23, 61, 40, 5, 1), (100, 151, 33, 14, 1),
                                                (30, 96, 300, 9, 1), (26, 37, 48, 12, 0), (70, 151, 20, 66, 1)])
def test_learn_over_the_used_rails_matches_the_plain_learn_step(Rp, T1, B, rails, ddqn):
    """flm_dqn_learn_grads_u8_rails (rails from `rails` on empty in every observation: computed over the used rails,
    the empty ones added in closed form) against flm_dqn_learn_grads_u8 on the same batch: the loss, and every gradient
    within 1e-5 of its tensor's largest (the first dense layer's weights separately under the used and under the empty
    rows); in two phases the same bits as in one call; next states behind the states and apart."""
    import types
    from flatland_marl_amd.agent import DQNAgent
    from flatland_marl_amd._lib import call, ptr, query
    ps = (Rp * T1 + 255) // 256 * 256
    H3 = 3 * Rp
    args = types.SimpleNamespace(prediction_depth=T1 - 1, lr=5e-5, buffer_size=10, batch_size=B, update_every=10,
                                 gamma=0.99, tau=1e-3)
    torch.manual_seed(5)
    agent = DQNAgent(args, action_space=2, bitmap_height=H3)
    with torch.no_grad():
        flat = agent.qnetwork_local.flat_params()
        flat.mul_(torch.exp(torch.randn(flat.shape, generator=torch.Generator().manual_seed(2))).to(flat.device))
    local, target = agent.qnetwork_local.flat_params(), agent.qnetwork_target.flat_params()
    target.copy_(local * 0.93)
    g = torch.Generator().manual_seed(B + rails)
    obs = torch.zeros((2 * B, 3, Rp, T1), dtype=torch.uint8)
    obs[:, 0, :rails] = torch.randint(-1, 2, (2 * B, rails, T1), generator=g).to(torch.int8).view(torch.uint8)
    obs[:, 1:, :rails] = torch.randint(0, 9, (2 * B, 2, rails, T1), generator=g).to(torch.uint8)
    both = torch.full((2 * B, 3, ps), 55, dtype=torch.uint8)
    both[:, :, :Rp * T1] = obs.view(2 * B, 3, Rp * T1)
    both = both.cuda()
    act = torch.randint(0, 2, (B,), generator=g).cuda()
    rew = torch.randn(B, generator=g).cuda()
    done = (torch.rand(B, generator=g) < 0.2).float().cuda()
    st = torch.cuda.current_stream().cuda_stream
    n = local.numel()
    ws = torch.empty(query("flm_dqn_learn_workspace_u8", B, Rp, T1, 2), device="cuda")
    head = lambda s, sn: (ptr(local), ptr(target), ptr(s), ptr(sn), ps, ptr(act), ptr(rew), ptr(done), 0.97, ddqn, B, Rp, T1, 2)
    want, loss_w = torch.zeros(n, device="cuda"), torch.zeros(1, device="cuda")
    call("flm_dqn_learn_grads_u8", *head(both[:B], both[B:]), ptr(ws), ptr(want), ptr(loss_w), st)
    got, loss_g = torch.zeros(n, device="cuda"), torch.zeros(1, device="cuda")
    call("flm_dqn_learn_grads_u8_rails", *head(both[:B], both[B:]), rails, ptr(ws), ptr(got), ptr(loss_g), 1, 0, st)
    torch.cuda.synchronize()
    assert torch.allclose(loss_g, loss_w, rtol=1e-5)
    shapes = [(64, T1), (64,), (512, 64, 3, Rp), (512,), (512, 64, 3, Rp), (512,), (2, 512), (2,), (1, 512), (1,)]
    off = 0
    R4 = (rails + 3) & ~3
    for shp in shapes:
        k = int(np.prod(shp))
        a, b = got[off:off + k].view(shp), want[off:off + k].view(shp)
        parts = [(a, b)] if len(shp) != 4 else [(a[..., :R4], b[..., :R4]), (a[..., R4:], b[..., R4:])]
        for x, y in parts:
            if y.numel():
                top = y.abs().max().item()
                assert (x - y).abs().max().item() <= 1e-5 * top + 1e-30, (off, shp, (x - y).abs().max().item(), top)
        off += k
    assert off == n and want.abs().max().item() > 0
    # two phases: the same bits as the one call
    two, loss_t = torch.full((n,), 7.0, device="cuda"), torch.zeros(1, device="cuda")
    call("flm_dqn_learn_grads_u8_rails", *head(both[:B], both[B:]), rails, ptr(ws), ptr(two), ptr(loss_t), 1, 1, st)
    call("flm_dqn_learn_grads_u8_rails", *head(both[:B], both[B:]), rails, ptr(ws), ptr(two), ptr(loss_t), 1, 2, st)
    # next states apart from the states: the same bits again
    s_apart, sn_apart = both[:B].clone(), both[B:].clone()
    apart, loss_a = torch.zeros(n, device="cuda"), torch.zeros(1, device="cuda")
    call("flm_dqn_learn_grads_u8_rails", *head(s_apart, sn_apart), rails, ptr(ws), ptr(apart), ptr(loss_a), 1, 0, st)
    torch.cuda.synchronize()
    assert torch.equal(two, got) and torch.equal(loss_t, loss_g)
    assert torch.equal(apart, got) and torch.equal(loss_a, loss_g)
    # the gradient under the empty rows left to the optimizer pass: the same parameters, moments and target bit for bit
    rails4 = query("flm_dqn_learn_rails", B, Rp, T1, rails)
    if rails4:
        m0 = torch.randn(n, generator=torch.Generator().manual_seed(3)).cuda() * 1e-3
        v0 = torch.rand(n, generator=torch.Generator().manual_seed(4)).cuda() * 1e-6
        outs = []
        for fused in (False, True):
            p_, t_, m_, v_ = local.clone(), target.clone(), m0.clone(), v0.clone()
            gbuf = torch.full((n,), float("nan"), device="cuda")            # whatever is left unwritten must not be read
            call("flm_dqn_learn_grads_u8_rails", ptr(p_), ptr(t_), ptr(both[:B]), ptr(both[B:]), ps, ptr(act), ptr(rew), ptr(done),
                 0.97, ddqn, B, Rp, T1, 2, rails, ptr(ws), ptr(gbuf), ptr(loss_g), 0 if fused else 1, 0, st)
            if fused:
                off = query("flm_dqn_learn_rails_rcb_offset", B, Rp, T1, 2)
                call("flm_dqn_adam_soft_update_rails", ptr(p_), ptr(gbuf), ptr(m_), ptr(v_), ptr(t_), n, 1e-3, 0.9, 0.999, 1e-8, 3,
                     0.01, Rp, T1, 2, rails4, ptr(ws[off:off + 64]), None, None, None, None, st)
            else:
                assert not torch.isnan(gbuf).any()
                call("flm_dqn_adam_soft_update", ptr(p_), ptr(gbuf), ptr(m_), ptr(v_), ptr(t_), n, 1e-3, 0.9, 0.999, 1e-8, 3, 0.01, st)
            torch.cuda.synchronize()
            outs.append((p_, t_, m_, v_))
        for x, y in zip(*outs):
            assert torch.equal(x, y)
        assert not torch.equal(outs[0][0], local)


def _pack_weight_blocks(g, Rp, T1, r4):
    """The flat gradient with the two weight blocks of the first dense layer in the packed form: [512][64*3*r4] (filter,
    plane, rail < r4) at the start of each block, the rest of the block never read (NaN)."""
    K1 = 64 * 3 * Rp
    a1_w = 64 * T1 + 64
    out = g.clone()
    for w in (a1_w, a1_w + 512 * K1 + 512):
        blk = g[w:w + 512 * K1].view(512, 64, 3, Rp)[:, :, :, :r4].reshape(-1)
        out[w:w + 512 * K1] = float("nan")
        out[w:w + blk.numel()] = blk
    return out


def test_learn_step_leaves_the_packed_gradient_the_optimizer_reads():
    """fill_padding = 2: the two weight blocks' gradient under the used rails, packed at the start of each block, has the
    bits of the scattered form (fill_padding = 0); everything outside the two blocks is the same."""
    from flatland_marl_amd._lib import call, ptr, query
    Rp, T1, B, rails = 70, 151, 64, 17
    ps = (Rp * T1 + 255) // 256 * 256
    n = query("flm_dqn_param_count", 3 * Rp, T1, 2)
    g = torch.Generator().manual_seed(21)
    local = (torch.randn(n, generator=g) * 0.05).cuda()
    target = (torch.randn(n, generator=g) * 0.05).cuda()
    both = torch.zeros((2 * B, 3, ps), dtype=torch.uint8)
    both[:, :, :rails * T1] = (torch.rand((2 * B, 3, rails * T1), generator=g) < 0.05).to(torch.uint8)
    both = both.cuda()
    act = torch.randint(0, 2, (B,), generator=g).cuda()
    rew, done = torch.randn(B, generator=g).cuda(), (torch.rand(B, generator=g) < 0.2).float().cuda()
    ws = torch.empty(query("flm_dqn_learn_workspace_u8", B, Rp, T1, 2), device="cuda")
    r4 = query("flm_dqn_learn_rails", B, Rp, T1, rails)
    assert r4 == 20
    outs = []
    for fill in (0, 2):
        gbuf, loss = torch.full((n,), float("nan"), device="cuda"), torch.zeros(1, device="cuda")
        call("flm_dqn_learn_grads_u8_rails", ptr(local), ptr(target), ptr(both[:B]), ptr(both[B:]), ps, ptr(act), ptr(rew),
             ptr(done), 0.97, 1, B, Rp, T1, 2, rails, ptr(ws), ptr(gbuf), ptr(loss), fill, 0, 0)
        torch.cuda.synchronize()
        outs.append((gbuf, loss))
    want = _pack_weight_blocks(outs[0][0], Rp, T1, r4)
    got = outs[1][0]
    K1, Kc = 64 * 3 * Rp, 64 * 3 * r4
    a1_w = 64 * T1 + 64
    for w in (a1_w, a1_w + 512 * K1 + 512):
        assert torch.equal(got[w:w + 512 * Kc], want[w:w + 512 * Kc]) and not torch.isnan(got[w:w + 512 * Kc]).any()
        assert got[w:w + 512 * Kc].abs().max() > 0
        got[w:w + 512 * K1] = 0
        want[w:w + 512 * K1] = 0
    assert torch.equal(got, want) and torch.equal(outs[0][1], outs[1][1])
    with pytest.raises(Exception):                      # a batch too small for the used-rails form has no packed form
        call("flm_dqn_learn_grads_u8_rails", ptr(local), ptr(target), ptr(both[:4]), ptr(both[B:B + 4]), ps, ptr(act), ptr(rew),
             ptr(done), 0.97, 1, 4, Rp, T1, 2, rails, ptr(ws), ptr(outs[0][0]), ptr(outs[0][1]), 2, 0, 0)


def test_optimizer_in_the_compact_form_gives_the_same_bits():
    """FlatAdam keeps its data of the first dense layer's weights without the empty rows while every step ran over the
    used rails: parameters and target after several steps (the used rails growing on the way), and the moments once
    they are read, equal those of the pass that keeps every moment."""
    from flatland_marl_amd._lib import call, ptr, query
    from flatland_marl_amd.agent import FlatAdam
    from flatland_marl_amd.model import Dueling_DQN
    Rp, T1, NA = 70, 151, 2
    torch.manual_seed(5)
    net = Dueling_DQN(T1, 3 * Rp, NA).to("cuda")
    flat = net.flat_params()
    n = flat.numel()
    assert n == query("flm_dqn_param_count", 3 * Rp, T1, NA)
    K1 = 64 * 3 * Rp
    a1_w = 64 * T1 + 64
    v1_w = a1_w + 512 * K1 + 512
    opt = FlatAdam(net, lr=1e-3)
    opt._ensure()
    p_e, t_e = flat.clone(), torch.randn(n, generator=torch.Generator().manual_seed(6)).cuda()
    t_c = t_e.clone()
    m_e, v_e = torch.zeros_like(flat), torch.zeros_like(flat)
    gen = torch.Generator().manual_seed(7)
    forms = []
    for step, r4 in enumerate((20, 20, 24, 24, 24, 32), 1):
        g = torch.randn(n, generator=gen).cuda() * 0.1
        for w in (a1_w, v1_w):                                   # what a learn step with fill_padding = 0 leaves unwritten
            g[w:w + 512 * K1].view(512, 64, 3, Rp)[:, :, :, r4:] = float("nan")
        rcb = torch.rand(64, generator=gen).cuda()
        rcb[::5] = 0.0
        call("flm_dqn_adam_soft_update_rails", ptr(p_e), ptr(g), ptr(m_e), ptr(v_e), ptr(t_e), n, 1e-3, 0.9, 0.999, 1e-8, step,
             0.01, Rp, T1, NA, r4, ptr(rcb), None, None, None, None, 0)
        assert opt.compact_for(Rp, T1, NA, r4)
        opt.grads.copy_(_pack_weight_blocks(g, Rp, T1, r4))
        opt.step(target=t_c, tau=0.01, rails=(Rp, T1, NA, r4, rcb, True))
        forms.append(opt._compact)
        torch.cuda.synchronize()
        assert torch.equal(flat, p_e) and torch.equal(t_c, t_e), step
    assert forms == [(Rp, T1, NA, 20)] * 2 + [(Rp, T1, NA, 24)] * 3 + [(Rp, T1, NA, 32)]
    assert not torch.equal(opt._m, m_e)                           # the large buffers were indeed left behind
    assert torch.equal(opt.m, m_e) and torch.equal(opt.v, v_e)    # reading them brings them up to date
    assert opt._compact is None and opt._uniform_from is None
    assert not torch.isnan(m_e).any() and not torch.isnan(p_e).any()
    # ... and from then on every moment is kept: one more step over the used rails, then a plain one
    g = torch.randn(n, generator=gen).cuda() * 0.1
    rcb = torch.rand(64, generator=gen).cuda()
    call("flm_dqn_adam_soft_update_rails", ptr(p_e), ptr(g), ptr(m_e), ptr(v_e), ptr(t_e), n, 1e-3, 0.9, 0.999, 1e-8, 7, 0.01,
         Rp, T1, NA, 32, ptr(rcb), None, None, None, None, 0)
    assert not opt.compact_for(Rp, T1, NA, 32)
    with pytest.raises(ValueError):
        opt.step(target=t_c, tau=0.01, rails=(Rp, T1, NA, 32, rcb, True))
    opt.step_count -= 1
    opt.grads.copy_(g)
    opt.step(target=t_c, tau=0.01, rails=(Rp, T1, NA, 32, rcb, False))
    assert opt._compact is None
    call("flm_dqn_adam_soft_update", ptr(p_e), ptr(g), ptr(m_e), ptr(v_e), ptr(t_e), n, 1e-3, 0.9, 0.999, 1e-8, 8, 0.01, 0)
    opt.step(target=t_c, tau=0.01)
    torch.cuda.synchronize()
    assert torch.equal(flat, p_e) and torch.equal(t_c, t_e) and torch.equal(opt._m, m_e) and torch.equal(opt._v, v_e)


def test_a_plain_step_after_steps_over_the_used_rails_finds_every_moment():
    """A step that is not over the used rails (here: the float learn path's) first spreads the compact moments."""
    from flatland_marl_amd._lib import call, ptr
    from flatland_marl_amd.agent import FlatAdam
    from flatland_marl_amd.model import Dueling_DQN
    Rp, T1, NA = 40, 31, 2
    torch.manual_seed(8)
    net = Dueling_DQN(T1, 3 * Rp, NA).to("cuda")
    flat = net.flat_params()
    n = flat.numel()
    opt = FlatAdam(net, lr=1e-3)
    opt._ensure()
    p_e, t_e = flat.clone(), torch.zeros_like(flat)
    t_c, m_e, v_e = t_e.clone(), torch.zeros_like(flat), torch.zeros_like(flat)
    gen = torch.Generator().manual_seed(9)
    for step in (1, 2, 3):
        g = torch.randn(n, generator=gen).cuda()
        rcb = torch.rand(64, generator=gen).cuda()
        if step < 3:
            call("flm_dqn_adam_soft_update_rails", ptr(p_e), ptr(g), ptr(m_e), ptr(v_e), ptr(t_e), n, 1e-3, 0.9, 0.999, 1e-8,
                 step, 0.01, Rp, T1, NA, 12, ptr(rcb), None, None, None, None, 0)
            opt.grads.copy_(_pack_weight_blocks(g, Rp, T1, 12))
        else:
            call("flm_dqn_adam_soft_update", ptr(p_e), ptr(g), ptr(m_e), ptr(v_e), ptr(t_e), n, 1e-3, 0.9, 0.999, 1e-8, step,
                 0.01, 0)
            opt.grads.copy_(g)
        opt.step(target=t_c, tau=0.01, rails=(Rp, T1, NA, 12, rcb, True) if step < 3 else None)
        assert (opt._compact is not None) == (step < 3)
    torch.cuda.synchronize()
    assert torch.equal(flat, p_e) and torch.equal(t_c, t_e) and torch.equal(opt._m, m_e) and torch.equal(opt._v, v_e)


def test_moments_round_trip_and_argument_checks_of_the_compact_form():
    """flm_dqn_adam_moments: taking the compact form from m and v and writing it back restores them bit for bit when the
    moments are equal along the empty rows; the entry points refuse pointer sets that are not all-or-none, the same
    buffer for moments_in and moments_out, and sizes that do not match the network."""
    from flatland_marl_amd._lib import FlmError, call, ptr, query
    Rp, T1, NA, r4 = 23, 31, 3, 8
    n = query("flm_dqn_param_count", 3 * Rp, T1, NA)
    K1 = 64 * 3 * Rp
    a1_w = 64 * T1 + 64
    v1_w = a1_w + 512 * K1 + 512
    g = torch.Generator().manual_seed(31)
    m, v = torch.randn(n, generator=g).cuda(), torch.rand(n, generator=g).cuda()
    for t in (m, v):                                             # equal along the empty rows of every (unit, filter)
        for w in (a1_w, v1_w):
            blk = t[w:w + 512 * K1].view(512, 64, 3, Rp)
            blk[:, :, :, r4:] = blk[:, :, 0:1, r4:r4 + 1]
    m0, v0 = m.clone(), v.clone()
    pairs = torch.empty((2 * 512 * 64, 2), device="cuda")
    mpk, vpk = (torch.empty(2 * 512 * 64 * 3 * r4, device="cuda") for _ in range(2))
    call("flm_dqn_adam_moments", ptr(m), ptr(v), n, Rp, T1, NA, r4, ptr(pairs), ptr(mpk), ptr(vpk), 0, 0)
    torch.cuda.synchronize()
    assert torch.equal(m, m0) and torch.equal(v, v0)             # taking leaves m and v alone
    want = m0[a1_w:a1_w + 512 * K1].view(512, 64, 3, Rp)[:, :, :, :r4].reshape(-1)
    assert torch.equal(mpk[:want.numel()], want)
    assert torch.equal(pairs[:512 * 64, 1], v0[a1_w:a1_w + 512 * K1].view(512, 64, 3, Rp)[:, :, 0, r4].reshape(-1))
    for w in (a1_w, v1_w):                                       # what is stale while the compact form is current
        m[w:w + 512 * K1] = float("nan")
        v[w:w + 512 * K1] = float("nan")
    call("flm_dqn_adam_moments", ptr(m), ptr(v), n, Rp, T1, NA, r4, ptr(pairs), ptr(mpk), ptr(vpk), 1, 0)
    torch.cuda.synchronize()
    assert torch.equal(m, m0) and torch.equal(v, v0)
    p, t, gr, rcb = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(64, device="cuda")
    pairs2 = torch.empty_like(pairs)
    head = (ptr(p), ptr(gr), ptr(m), ptr(v), ptr(t), n, 1e-3, 0.9, 0.999, 1e-8, 1, 0.01, Rp, T1, NA, r4, ptr(rcb))
    for extra in ((ptr(pairs), None, ptr(mpk), ptr(vpk)), (ptr(pairs), ptr(pairs2), None, ptr(vpk)),
                  (ptr(pairs), ptr(pairs), ptr(mpk), ptr(vpk)), (None, ptr(pairs2), None, None)):
        with pytest.raises(FlmError):
            call("flm_dqn_adam_soft_update_rails", *head, *extra, 0)
    with pytest.raises(FlmError):                                # rails4 must be a multiple of four below max_rails
        call("flm_dqn_adam_soft_update_rails", *head[:15], 6, ptr(rcb), None, None, None, None, 0)
    with pytest.raises(FlmError):                                # n of another network
        call("flm_dqn_adam_soft_update_rails", *head[:5], n - 1, *head[6:], None, None, None, None, 0)
    with pytest.raises(FlmError):
        call("flm_dqn_adam_moments", ptr(m), ptr(v), n, Rp, T1, NA, Rp, ptr(pairs), ptr(mpk), ptr(vpk), 0, 0)
    call("flm_dqn_adam_soft_update_rails", *head, ptr(pairs), ptr(pairs2), ptr(mpk), ptr(vpk), 0)   # and a well-formed call passes
    torch.cuda.synchronize()
