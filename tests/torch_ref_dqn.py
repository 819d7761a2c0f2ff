"""Plain PyTorch fp32 statement of the network and of one learn step, used as the checker of the HIP
DQN kernels (architecture of /root/reference/src/model.py:11-43, update of src/agent.py:94-131)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class TorchDuelingDQN(nn.Module):
    def __init__(self, width, height, action_space=2):
        super().__init__()
        self.conv1 = nn.Conv2d(1, 64, kernel_size=(1, width))
        self.fc1_adv = nn.Linear(64 * height, 512)
        self.fc1_val = nn.Linear(64 * height, 512)
        self.fc2_adv = nn.Linear(512, action_space)
        self.fc2_val = nn.Linear(512, 1)

    def forward(self, x):
        x = F.relu(self.conv1(x)).flatten(1)
        adv = self.fc2_adv(F.relu(self.fc1_adv(x)))
        val = self.fc2_val(F.relu(self.fc1_val(x)))
        return val + adv - adv.mean(1, keepdim=True)


def learn_step(local, target, optimizer, batch, gamma, tau):
    s, a, r, sn, d = batch
    q = local(s).gather(1, a.unsqueeze(-1)).squeeze(-1)
    best = local(sn).max(1)[1]
    q_next = target(sn).gather(1, best.unsqueeze(-1)).squeeze(-1).detach()
    loss = F.mse_loss(q, r + gamma * q_next * (1 - d))
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    with torch.no_grad():
        for tp, lp in zip(target.parameters(), local.parameters()):
            tp.copy_(tau * lp + (1.0 - tau) * tp)
    return loss
