"""world_size-2 gloo run of the sharding / timing / gradient-averaging helpers (CPU)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from flatland_marl_amd import parallel
    from flatland_marl_amd.synthetic import make_batch
    import torch.distributed as dist
    r, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    lo, hi = parallel.shard_range(n_total, r, w)
    # every rank builds only its own environments; seeds are global environment indices
    fx = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "envs", "Test_0_Level_0.npz")))
    mine = make_batch([fx], hi - lo, 5, seed=11, square_only=True, first=lo)
    t = parallel.max_over_ranks(1.0 + rank)
    total = parallel.sum_over_ranks(hi - lo)
    fewest = parallel.min_over_ranks(3 + 2 * rank)
    g = torch.full((1000,), float(rank + 1))
    parallel.allreduce_mean_(g)
    # the learn step's exchange in two parts: both are started before either is waited for
    g2 = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    ex = parallel.GradientExchange()
    ex.begin(g2[100:])
    ex.begin(g2[:100])
    ex.finish()
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), lo=lo, hi=hi, t=t, total=total, fewest=fewest, g=g.numpy(),
             g2=g2.numpy(), g2_parts=np.array([n for _, n in ex.parts]),
             start=mine["start"], target=mine["target"], grid=mine["grid"], speed=mine["speed"],
             direction=mine["direction"])
    dist.destroy_process_group()


def test_two_ranks_shard_and_reduce(tmp_path):
    world, n_total = 2, 7
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    r = [dict(np.load(os.path.join(tmp_path, "r%d.npz" % i))) for i in range(world)]
    assert int(r[0]["lo"]) == 0 and int(r[0]["hi"]) == int(r[1]["lo"]) and int(r[1]["hi"]) == n_total
    assert float(r[0]["t"]) == float(r[1]["t"]) == 2.0           # max over ranks
    assert float(r[0]["total"]) == float(r[1]["total"]) == n_total
    assert int(r[0]["fewest"]) == int(r[1]["fewest"]) == 3          # learn steps every rank can take together
    assert np.allclose(r[0]["g"], 1.5) and np.allclose(r[1]["g"], 1.5)
    for i in range(world):
        assert np.array_equal(r[i]["g2"], np.arange(1000, dtype=np.float32) * 1.5) and r[i]["g2_parts"].tolist() == [900, 100]
    # shards are what a single process would have built for the same global environment indices
    from flatland_marl_amd.synthetic import make_batch
    fx = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "envs", "Test_0_Level_0.npz")))
    whole = make_batch([fx], n_total, 5, seed=11)
    for i in range(world):
        lo, hi = int(r[i]["lo"]), int(r[i]["hi"])
        for key in ("grid", "start", "direction", "target", "speed"):
            assert np.array_equal(r[i][key], whole[key][lo:hi]), (i, key)
    assert not np.array_equal(r[0]["start"][0], r[1]["start"][0])     # the ranks hold different environments


def test_shard_range_tiles():
    from flatland_marl_amd.parallel import shard_range
    for n in (1, 7, 8, 4096, 4099):
        for w in (1, 2, 3, 8):
            edges = [shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
