"""Where the time of flm_altpaths goes (development tool): the requests of a rollout step some way into an episode of
the bench's 1024 x 10 workload, timed with the descents cut short (max_depth) and with fewer requests."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import flatland_marl_amd as fm  # noqa: E402
from flatland_marl_amd._lib import call, ptr  # noqa: E402
from flatland_marl_amd.agent import DQNAgent  # noqa: E402
from flatland_marl_amd.rollout import RolloutBatch  # noqa: E402
from flatland_marl_amd.train import BatchedTrainer  # noqa: E402


def main():
    torch.manual_seed(0)
    data, E, A, depth, _ = bench.workload_batch("cfg2g", 0)
    b = fm.ObservationBatch(data["grid"], data["start"], data["direction"], data["target"], data["speed"], depth, max_rails=70)
    ro = RolloutBatch(b, data["speed"], data["direction"], max_alt_depth=203,
                      malfunction=dict(rate=2000, min_duration=0, max_duration=0, seed=0)).reset()
    a = types.SimpleNamespace(prediction_depth=depth, lr=5e-5, buffer_size=20000, batch_size=512, update_every=10,
                              gamma=0.99, tau=1e-3)
    tr = BatchedTrainer(ro, DQNAgent(a, action_space=2, bitmap_height=3 * b.Rp), start_eps=0.1)
    for _ in range(30):
        tr.step(train=False)
    # the requests of the next step, as RolloutBatch.step forms them
    ro.classify()
    call("flm_rollout_requests", ptr(ro.branch), ptr(ro.status), ptr(ro.pos), ptr(ro.heading), ptr(b.start_cell), ro.E, ro.A,
         ptr(ro.ready_list), ptr(ro.switch_list), ptr(ro.req_counts), ptr(ro.alt_slot), ptr(ro._net), ptr(ro._best),
         ptr(b.pos_cell), ptr(b.direction), ptr(b.status), ptr(ro._req_ws), ro.stream)
    n_ready, n_switch = ro.read_counts()[:2]
    req = ro.switch_list[:n_switch].clone()
    print("requests", n_switch, "departing", n_ready, "cap_cells", b.cap_cells, "Lp", ro.Lp)
    K, dev = ro.cap_alt, ro.device
    i32 = dict(dtype=torch.int32, device=dev)
    out = dict(count=torch.empty((n_switch,), **i32), len=torch.zeros((n_switch, K), **i32),
               cell=torch.empty((n_switch, K, ro.Lp), **i32), meta=torch.empty((n_switch, K, ro.Lp), **i32),
               err=torch.empty((n_switch,), **i32))

    def run(n, max_depth):
        call("flm_altpaths", ptr(b.grid), ptr(b.cell_idx), ptr(b.dist), ptr(b.succ_table()), ptr(b.choice_table()),
             ptr(b.idx_cell), ptr(b.tslot), ptr(b.node_id), ptr(req), n, ptr(b.pos_cell), ptr(b.direction), ptr(b.target_cell), ro.E, ro.A,
             b.H, b.W, b.cap_cells, max_depth, K, ro.Lp, ptr(out["count"]), ptr(out["len"]), ptr(out["cell"]),
             ptr(out["meta"]), ptr(out["err"]), ro.stream)

    def timed(n, max_depth, reps=20):
        for _ in range(3):
            run(n, max_depth)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for x, y in ev:
            x.record()
            run(n, max_depth)
            y.record()
        torch.cuda.synchronize()
        return float(np.median([x.elapsed_time(y) for x, y in ev])) * 1e3

    for n in (1, 64, 256, n_switch):
        print("n_req %5d:" % n, "  ".join("depth %3d %7.1f us" % (d, timed(n, d)) for d in (0, 1, 8, 32, 100, 203)))
    run(n_switch, 203)
    torch.cuda.synchronize()
    ln = out["len"].cpu().numpy()
    cnt = out["count"].cpu().numpy()
    print("alternatives per request: mean %.2f max %d; path elements: mean %.1f max %d" %
          (cnt.mean(), cnt.max(), ln[ln > 0].mean(), ln.max()))


if __name__ == "__main__":
    main()
