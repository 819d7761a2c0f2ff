"""Data-parallel training rehearsed with two ranks on one GPU (gloo carries the gradient exchange here; on a
multi-GPU node the same code path runs over RCCL): different environments per rank, identical weights
after every learn step."""
import os
import socket
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      FLM_DIST_BACKEND="gloo")
    import torch
    import torch.distributed as dist
    import flatland_marl_amd as fm
    from flatland_marl_amd import parallel
    from flatland_marl_amd.agent import DQNAgent
    from flatland_marl_amd.rollout import RolloutBatch
    from flatland_marl_amd.synthetic import make_batch
    from flatland_marl_amd.train import BatchedTrainer
    device = torch.device("cuda", 0)
    parallel.init_distributed(None, device)
    # another map per rank, with another number of rails: the ranks have to agree on the rows they leave to the closed forms
    fx = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "envs",
                                   ("Test_0_Level_0.npz", "Test_0_Level_1.npz")[rank % 2])))
    lo, hi = parallel.shard_range(6, rank, world)
    data = make_batch([fx], hi - lo, 8, seed=100 + lo)          # every rank steps its own environments
    depth, Rp = 40, 40
    b = fm.ObservationBatch(data["grid"], data["start"], data["direction"], data["target"], data["speed"], depth,
                            max_rails=Rp)
    ro = RolloutBatch(b, data["speed"], data["direction"]).reset()
    args = types.SimpleNamespace(prediction_depth=depth, lr=1e-3, buffer_size=500, batch_size=16, update_every=4,
                                 gamma=0.99, tau=1e-2)
    torch.manual_seed(0)                                          # identical initial weights on every rank
    agent = DQNAgent(args, action_space=2, bitmap_height=3 * Rp)
    w0 = agent.qnetwork_local.flat_params().clone()
    tr = BatchedTrainer(ro, agent, start_eps=0.3, max_learns_per_step=2, max_steps=40, seed=rank)
    out = tr.run_episode(train=True)
    torch.cuda.synchronize()
    w = agent.qnetwork_local.flat_params()
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), w=w.cpu().numpy(), changed=float((w - w0).abs().max().item()),
             learns=out["learns"], transitions=out["transitions"], steps=out["steps"],
             target=agent.qnetwork_target.flat_params().cpu().numpy(), start=data["start"], goal=data["target"],
             final_pos=ro.pos.cpu().numpy(), flagged=out["agents_flagged"],
             parts=np.array([n for _, n in agent.exchange.parts], np.int64),
             part_offsets=np.array([(p - agent.optimizer.grads.data_ptr()) // 4 for p, _ in agent.exchange.parts], np.int64),
             n_params=w.numel(), rails4=agent.optimizer._compact[3] if agent.optimizer._compact else 0,
             own_rails=ro.rails_used())
    dist.destroy_process_group()


def test_two_ranks_keep_identical_weights(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [dict(np.load(os.path.join(tmp_path, "r%d.npz" % i))) for i in range(world)]
    assert int(r[0]["learns"]) == int(r[1]["learns"]) > 0          # ranks agree on the number of learn steps
    assert int(r[0]["steps"]) == int(r[1]["steps"])                 # and on when the episode ends
    # the ranks really stepped different environments: other agents, other targets, other final positions
    assert not np.array_equal(r[0]["start"], r[1]["start"]) and not np.array_equal(r[0]["goal"], r[1]["goal"])
    assert not np.array_equal(r[0]["final_pos"], r[1]["final_pos"])
    assert int(r[0]["transitions"]) > 0 and int(r[1]["transitions"]) > 0
    assert float(r[0]["changed"]) > 0
    # the exchange went in two parts that tile the flat gradient: dense layers first (while the convolution's
    # gradients were still being computed), then the convolution's
    # (the last learn step ran over the used rails with the optimizer in its compact form: of the two large weight blocks
    # only the packed gradient at their start travels - three runs instead of one)
    own = [int(r[i]["own_rails"]) for i in range(world)]
    assert own[0] != own[1] and int(r[0]["rails4"]) == int(r[1]["rails4"]) == (max(own) + 3) // 4 * 4
    for i in range(world):
        n, conv, block = int(r[i]["n_params"]), 64 * 41 + 64, 512 * 64 * 120
        cols = 64 * 3 * int(r[i]["rails4"])
        assert 0 < cols < 64 * 120
        assert r[i]["part_offsets"].tolist() == [conv, conv + block, conv + 2 * block + 512, 0]
        assert r[i]["parts"].tolist() == [512 * cols, 512 + 512 * cols, n - (conv + 2 * block + 512), conv]
    assert np.array_equal(r[0]["w"], r[1]["w"])                     # same averaged gradient, same Adam state
    assert np.array_equal(r[0]["target"], r[1]["target"])
