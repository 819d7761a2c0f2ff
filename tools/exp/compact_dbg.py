import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import flatland_marl_amd as fm
from flatland_marl_amd.agent import DQNAgent
from flatland_marl_amd.rollout import RolloutBatch
from flatland_marl_amd.train import BatchedTrainer
torch.manual_seed(0)
data, E, A, depth, _ = bench.workload_batch("cfg2g", 0)
b = fm.ObservationBatch(data["grid"], data["start"], data["direction"], data["target"], data["speed"], depth, max_rails=70)
ro = RolloutBatch(b, data["speed"], data["direction"], max_alt_depth=203, malfunction=dict(rate=2000, min_duration=0, max_duration=0, seed=0)).reset()
a = types.SimpleNamespace(prediction_depth=depth, lr=5e-5, buffer_size=20000, batch_size=512, update_every=10, gamma=0.99, tau=1e-3)
tr = BatchedTrainer(ro, DQNAgent(a, action_space=2, bitmap_height=3 * b.Rp), start_eps=0.1, max_learns_per_step=4)
opt = tr.agent.optimizer if hasattr(tr, "agent") else None
for i in range(8):
    tr.step(train=True)
    o = tr.policy.agent.optimizer if opt is None else opt
    print(i, "compact", o._compact, "uniform_from", o._uniform_from, "steps", o.step_count, "rails_seen", getattr(tr, "_rails_seen", None))
