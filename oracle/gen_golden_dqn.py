"""Golden vectors of the reference's own Dueling_DQN / DQNAgent (build container only).

TEST INFRASTRUCTURE ONLY.  Weights are not stored (78.7 MB): they are torch's default initialisation
under a fixed seed, which the tests reproduce with the same torch version.

    python oracle/gen_golden_dqn.py
"""
import argparse
import collections
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


if __name__ == "__main__":
    if "plain" in sys.argv[1:]:
        main_plain()
    else:
        main()
