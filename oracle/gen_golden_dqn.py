"""Golden vectors of the reference's own Dueling_DQN / DQNAgent (build container only).

TEST INFRASTRUCTURE ONLY.  Weights are not stored (78.7 MB): they are torch's default initialisation
under a fixed seed, which the tests reproduce with the same torch version.

    python oracle/gen_golden_dqn.py
"""
import argparse
import collections
import copy
import collections.abc
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
collections.Iterable = collections.abc.Iterable      # src/agent.py:4 imports it from `collections`

from src.model import Dueling_DQN  # noqa: E402
from src.agent import DQNAgent  # noqa: E402
from tests import dqn_inputs  # noqa: E402


def tensor_sums(module):
    return np.array([[float(p.double().sum()), float(p.double().abs().sum())] for p in module.parameters()])


def main():
    out = {}
    H3, T1 = 3 * dqn_inputs.MAX_RAILS, dqn_inputs.DEPTH + 1
    # ---- forward
    torch.manual_seed(1234)
    net = Dueling_DQN(T1, H3, 2)
    x = torch.from_numpy(dqn_inputs.observations(6))[:, None]
    with torch.no_grad():
        out["act_q32"] = net(x).numpy()
        out["act_q64"] = net.double()(x.double()).numpy()
    out["act_param_sums"] = tensor_sums(net)
    # ---- one learn step
    B = 16
    args = argparse.Namespace(prediction_depth=dqn_inputs.DEPTH, lr=0.00005, buffer_size=100, batch_size=B,
                              update_every=10, gamma=0.99, tau=1e-3)
    torch.manual_seed(4321)
    agent = DQNAgent(args, action_space=2, bitmap_height=H3)
    s, a, r, sn, d = dqn_inputs.learn_batch(B)
    st = torch.from_numpy(s)[:, None]
    snt = torch.from_numpy(sn)[:, None]
    with torch.no_grad():
        out["learn_q_before"] = agent.qnetwork_local(st).numpy()
    exp = (st, torch.from_numpy(a), torch.from_numpy(r), snt, torch.from_numpy(d.astype(np.uint8)).float())
    agent.learn(exp, args.gamma)
    out["learn_grad_sums"] = np.array([[float(p.grad.double().sum()), float(p.grad.double().abs().sum())]
                                       for p in agent.qnetwork_local.parameters()])
    with torch.no_grad():
        out["learn_q_local_after"] = agent.qnetwork_local(st).numpy()
        out["learn_q_target_after"] = agent.qnetwork_target(st).numpy()
    out["learn_local_sums"] = tensor_sums(agent.qnetwork_local)
    out["learn_target_sums"] = tensor_sums(agent.qnetwork_target)
    # a second step exercises the Adam moments
    agent.learn(exp, args.gamma)
    with torch.no_grad():
        out["learn2_q_local_after"] = agent.qnetwork_local(st).numpy()
    out["learn2_local_sums"] = tensor_sums(agent.qnetwork_local)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dqn_reference.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.ravel()[:4])


def main_plain():
    """One learn step of the reference's agent with double_dqn=False (src/agent.py:103-105) -> dqn_reference_plain.npz.
    Its Q_targets_next keeps a trailing axis of length one, so the targets broadcast to [B,B] against the [B]
    expected values inside mse_loss (torch warns about the sizes): what is recorded here is that behaviour."""
    import warnings
    out = {}
    H3 = 3 * dqn_inputs.MAX_RAILS
    B = 16
    args = argparse.Namespace(prediction_depth=dqn_inputs.DEPTH, lr=0.00005, buffer_size=100, batch_size=B,
                              update_every=10, gamma=0.99, tau=1e-3)
    torch.manual_seed(4321)
    agent = DQNAgent(args, action_space=2, bitmap_height=H3, double_dqn=False)
    with torch.no_grad():                       # a target network that differs from the local one
        for p in agent.qnetwork_target.parameters():
            p.mul_(1.05)
    s, a, r, sn, d = dqn_inputs.learn_batch(B)
    st = torch.from_numpy(s)[:, None]
    snt = torch.from_numpy(sn)[:, None]
    exp = (st, torch.from_numpy(a), torch.from_numpy(r), snt, torch.from_numpy(d.astype(np.uint8)).float())
    losses = []
    import torch.nn.functional as F
    real = F.mse_loss

    def spy(x, y, *aa, **kw):
        v = real(x, y, *aa, **kw)
        losses.append(float(v))
        return v
    F.mse_loss = spy
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        agent.learn(exp, args.gamma)
    F.mse_loss = real
    out["loss"] = np.array(losses[0])
    out["grad_sums"] = np.array([[float(p.grad.double().sum()), float(p.grad.double().abs().sum())]
                                 for p in agent.qnetwork_local.parameters()])
    with torch.no_grad():
        out["q_local_after"] = agent.qnetwork_local(st).numpy()
        out["q_target_after"] = agent.qnetwork_target(st).numpy()
    out["local_sums"] = tensor_sums(agent.qnetwork_local)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dqn_reference_plain.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.ravel()[:4])


def main_trained():
    """Q-values of the reference's network after training has moved its weights -> dqn_reference_trained.npz.
    The reference's own DQNAgent runs 30 learn steps (lr 1e-3) on a fixed batch; the resulting weights are rounded to
    float16 so that they can be stored (2.7 MB) and loaded bit for bit on both sides, loaded back into the reference's
    Dueling_DQN, and its Q-values for 64 real observations are recorded in float32 and float64.  The network is the
    real architecture at max_rails = 7 (observations of bundled maps cropped to their first seven rails per plane):
    the weights of the full-size network would be 39 MB even as float16."""
    from tests import dqn_inputs as di
    Rp, T1 = di.TRAINED_RAILS, di.DEPTH + 1
    H3 = 3 * Rp
    B = 16
    args = argparse.Namespace(prediction_depth=di.DEPTH, lr=1e-3, buffer_size=100, batch_size=B, update_every=10, gamma=0.99,
                              tau=1e-3)
    torch.manual_seed(99)
    agent = DQNAgent(args, action_space=2, bitmap_height=H3)
    init = [p.detach().clone() for p in agent.qnetwork_local.parameters()]
    s, a, r, sn, d = di.trained_learn_batch(B)
    exp = (torch.from_numpy(s)[:, None], torch.from_numpy(a), torch.from_numpy(r), torch.from_numpy(sn)[:, None],
           torch.from_numpy(d.astype(np.uint8)).float())
    import torch.nn.functional as F
    real, losses = F.mse_loss, []

    def spy(x, y, *aa, **kw):
        v = real(x, y, *aa, **kw)
        losses.append(float(v))
        return v
    F.mse_loss = spy
    for _ in range(30):
        agent.learn(exp, args.gamma)
    F.mse_loss = real
    out = {"losses": np.array(losses)}
    moved = [float((p.detach() - p0).abs().max()) for p, p0 in zip(agent.qnetwork_local.parameters(), init)]
    out["moved_max"] = np.array(moved)
    # the stored weights: float16 roundings of the trained ones
    half = [p.detach().to(torch.float16) for p in agent.qnetwork_local.parameters()]
    for i, h in enumerate(half):
        out["w%d" % i] = h.numpy().view(np.uint16)
    net = Dueling_DQN(T1, H3, 2)
    with torch.no_grad():
        for p, h in zip(net.parameters(), half):
            p.copy_(h.float())
    x = torch.from_numpy(di.trained_observations(64))[:, None]
    with torch.no_grad():
        out["q32"] = net(x).numpy()
        out["q64"] = copy.deepcopy(net).double()(x.double()).numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dqn_reference_trained.npz"), **out)
    print("losses", losses[0], losses[-1], "moved", moved, "q", out["q64"][:3], "bytes",
          os.path.getsize(os.path.join(ROOT, "tests", "golden", "dqn_reference_trained.npz")))


if __name__ == "__main__":
    if "trained" in sys.argv[1:]:
        main_trained()
    elif "plain" in sys.argv[1:]:
        main_plain()
    else:
        main()
