"""Throughput of the observation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg5] [--envs E]

One step = one pass of the whole path over one resident batch of environments:
switch-graph build -> state numbering -> distance maps per distinct target ->
greedy shortest paths -> occupancy in column form -> observations (own plane +
two heat planes for every agent; the own plane is the dense occupancy plane).  Inputs (grids, agent states) are in HBM before the
timed region; nothing is copied to the host inside it.  With N > 1 every rank
runs its own shard of environments (no collective on the data path), so the
scaling is weak: per-GPU work is fixed.

Prints one JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (fixtures, envs per GPU, agents, depth, description)
    "cfg3": (["Test_8_Level_0", "Test_8_Level_1"], 4096, 50, 150,
             "4096 envs x 50 agents, 100x100 grid (Test_8-class maps, 8 symmetries x 2 levels, agents re-drawn), depth 150"),
    "cfg5": (["Test_9_Level_0", "Test_9_Level_1"], 256, 200, 300,
             "256 envs x 200 agents, 150x150 grid (Test_9-class maps), depth 300"),
    "cfg2": (["Test_2_Level_0", "Test_2_Level_1"], 1024, 10, 150,
             "1024 envs x 10 agents, 35x35 grid (Test_2-class maps stand in for 25x25), max_rails 70+, depth 150"),
    # the headline shape on 256 distinct generated 100x100 grids (about 580 rails each), so that the graph, chain and
    # distance kernels do not see the same few grids over and over
    "cfg3g": (dict(H=100, W=100, n_maps=256, spacing=5, keep=0.85, symmetries=False, scc_targets=True), 4096, 50, 150,
              "4096 envs x 50 agents, 256 distinct generated 100x100 grids (about 580 rails), agents re-drawn, depth 150"),
    # the same shape on grids from the package's own generator (flatland's generator is not available)
    "cfg2g": (dict(H=25, W=25, n_maps=8, spacing=4, max_rails=70), 1024, 10, 150,
              "1024 envs x 10 agents, generated 25x25 grids (8 grids x 8 symmetries), max_rails 70, depth 150"),
}


def load_fixtures(names):
    return [dict(np.load(os.path.join(ROOT, "tests", "golden", "envs", n + ".npz"))) for n in names]


def workload_batch(name, E=0, seed=0):
    """-> (data, E, A, depth, description) of a named workload; E = 0 takes the workload's own size."""
    from flatland_marl_amd.synthetic import make_batch
    from flatland_marl_amd.railgen import make_generated_batch
    src, E0, A, depth, desc = WORKLOADS[name]
    E = E or E0
    if isinstance(src, dict):
        data = make_generated_batch(E, A, seed=seed, depth=depth, **src)
    else:
        data = make_batch(load_fixtures(src), E, A, seed=seed)
    return data, E, A, depth, desc


def cpu_baseline(data, depth, Rp, budget_s=12.0, max_envs=64):
    """The CPU checker (a C restatement of the reference algorithm, one thread) on the first few
    environments of the same batch: graph + distance maps + paths + planes + get_obs for every agent."""
    from oracle import oracle as orc
    orc.lib()
    t0 = time.perf_counter()
    n_env = 0
    n_obs = 0
    E, A = data["start"].shape[:2]
    while n_env < min(E, max_envs) and (time.perf_counter() - t0) < budget_s:
        e = n_env
        fx = dict(grid=data["grid"][e], initial_position=data["start"][e], direction=data["direction"][e],
                  target=data["target"][e], speed=data["speed"][e], handle=np.arange(A))
        env = orc.OracleEnv(fx, depth)
        maps, _, _ = env.get_many()
        for a in range(A):
            orc.get_obs(maps, a, maps[a], Rp)
        n_env += 1
        n_obs += A
    dt = time.perf_counter() - t0
    out = dict(value=n_obs / dt, unit="agent-observations/s", cores=1, kind="port",
               sample="%d environments (%d agent-observations) of the same batch in %.1f s, single thread, "
                      "C restatement of the reference algorithm (oracle/flmarl_oracle.c)" % (n_env, n_obs, dt))
    ref = reference_python_constant()
    if ref is not None:
        out["reference_python"] = ref
    return out


def reference_python_constant():
    """The reference's own Python (src/rail_observations.py get_many + src/preprocessing.py get_obs for every
    agent, unmodified modules, one thread) timed once in the build container on the bundled 100x100 map this
    workload is drawn from; the reference cannot run beside the GPU, so the figure is a stated constant read
    from a committed file, not something this run measured."""
    path = os.path.join(ROOT, "tests", "golden", "ref_timing.json")
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    k = t.get("Test_8_Level_0_d150")
    if not k:
        return None
    secs = k["graph_s"] + k["dist_shim_s"] + k["get_many_s"] + k["get_obs_all_s"]
    return dict(value=k["agents"] / secs, unit="agent-observations/s", cores=1,
                host="build container, %d-core Xeon, Python 3.10, numpy 2.2" % t.get("_host", {}).get("cpus", 8),
                note="constant from tests/golden/ref_timing.json: Test_8/Level_0, %d agents, %d rails, depth 150: "
                     "graph %.2f s + distance map %.2f s + get_many %.2f s + get_obs for all agents %.2f s "
                     "(src/preprocessing.py:24-30 is the last term)"
                     % (k["agents"], k["rails"], k["graph_s"], k["dist_shim_s"], k["get_many_s"], k["get_obs_all_s"]))


def rollout_sample(device, steps=40, workload="cfg2g", max_rails=70, rank=0, world=1, envs=0, ref_cadence_steps=10):
    """Environment steps per second of the batched rollout (src/main.py:142-299 for every environment at once)
    on the 1024 x 10 workload (generated 25x25 grids, max_rails = 70; the script itself pads to 100 rails,
    src/main.py:91, which --rollout-max-rails 100 reproduces): the network
    consulted by every departing agent and every agent in front of a switch, first without, then with
    learning in two forms.  `act_and_learn_ref_cadence` is the reference's rule: one learn step per `update_every`
    = 10 stored transitions (src/agent.py:57-67, src/main.py:386-391), however many that is per rollout step; it is
    what a "full training rollout" means.  `act_and_learn` caps the learn steps of one rollout step at 4: a
    different, cheaper algorithm, kept as the figure of earlier runs.  Both say so in `learn_cadence`.  With
    several ranks every rank steps environments of its own and the learn steps all-reduce the gradient; value =
    all ranks' environment steps / the slowest rank's time."""
    import types
    import flatland_marl_amd as fm
    from flatland_marl_amd.agent import DQNAgent
    from flatland_marl_amd.rollout import RolloutBatch
    from flatland_marl_amd.train import BatchedTrainer
    from flatland_marl_amd import parallel
    data, E, A, depth, desc = workload_batch(workload, envs, seed=1000003 * rank)
    b = fm.ObservationBatch(data["grid"], data["start"], data["direction"], data["target"], data["speed"], depth,
                            max_rails=max_rails, device=device)
    # episodes of <= 200 steps (src/main.py:93); malfunction process with the script's defaults (src/main.py:373-375)
    malf = dict(rate=2000, min_duration=0, max_duration=0, seed=0)
    ro = RolloutBatch(b, data["speed"], data["direction"], max_alt_depth=max(200, depth + 1) + 2, malfunction=malf).reset()
    a = types.SimpleNamespace(prediction_depth=depth, lr=5e-5, buffer_size=20000, batch_size=512, update_every=10,
                              gamma=0.99, tau=1e-3)
    torch.manual_seed(0)                                     # identical replicas on every rank
    agent = DQNAgent(a, action_space=2, bitmap_height=3 * b.Rp)
    tr = BatchedTrainer(ro, agent, start_eps=0.1, max_learns_per_step=4, seed=rank)
    ref_steps = max(2, min(steps, ref_cadence_steps))
    out = {"unit": "env-steps/s", "config": {"workload": workload + ": " + desc, "envs_per_gpu": E, "n_gpus": world,
                                             "agents": A, "rails_padded": int(b.Rp),
                                             "rails_used": int(ro.rails_used()),
                                             "rails_note": "the environments of this workload have at most rails_used rails; "
                                                           "the rows up to rails_padded are the reference's padding to max_rails "
                                                           "(src/preprocessing.py:36-50): the network's kernels compute over the "
                                                           "used rails and add the padding's share in closed form "
                                                           "(flm_dqn_forward_u8_rails, flm_dqn_learn_grads_u8_rails), same "
                                                           "results up to fp32 rounding",
                                             "depth": depth, "steps_timed": steps, "eps": 0.1,
                                             "speeds": "1, 1/2, 1/3, 1/4", "malfunction": malf}}
    for key, train, cap, steps in (("act_only", False, 4, steps), ("act_and_learn", True, 4, steps),
                                   ("act_and_learn_ref_cadence", True, None, ref_steps)):
        tr.max_learns_per_step = cap
        ro.reset()
        for _ in range(6 if cap is not None else 3):         # first calls size the buffers
            tr.step(train=train)
        n0, l0 = tr.n_transitions, tr.n_learn
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(train=train)
        torch.cuda.synchronize()
        dt = parallel.max_over_ranks(time.perf_counter() - t0, device) / steps
        flagged, overflowed = ro.flagged_agents()
        out[key] = {"value": E * world / dt, "ms_per_step": dt * 1e3,
                    "transitions_per_step": (tr.n_transitions - n0) / steps,
                    "learn_steps_per_step": (tr.n_learn - l0) / steps,
                    "agents_flagged": flagged, "agents_over_capacity": overflowed, "steps_timed": steps}
        if train and tr.n_learn > l0:
            # the whole step over its learn steps (how many a step takes moves with what the policy has learnt)
            out[key]["ms_per_learn_step_incl_rollout"] = dt * 1e3 * steps / (tr.n_learn - l0)
        if train:
            out[key]["learn_cadence"] = {
                "update_every": a.update_every, "batch_size": a.batch_size, "cap_per_rollout_step": cap,
                "is_reference_cadence": cap is None,
                "note": "one learn step per update_every stored transitions (src/agent.py:57-67)" if cap is None else
                        "NOT the reference's cadence: learn steps per rollout step capped at %d" % cap}
        if overflowed:
            raise OverflowError("%d agents exceeded the rollout's alternative / path capacities" % overflowed)
    return out


def guarded_rollout_sample(out, rank, timeout_s, **kw):
    """Runs rollout_sample under a watchdog: the main JSON line must come out even if the sample stalls (a
    collective that one rank never enters).  On expiry rank 0 prints the line without the sample and every
    rank leaves with exit status 3; an exception in the sample is reported in the line and makes main()
    return 1."""
    import threading

    def expire():
        if rank == 0:
            out["env_steps"] = {"error": "no result within %d s" % timeout_s}
            print(json.dumps(out), flush=True)
        os._exit(3)                    # a stalled sample is a failed run: the line above keeps the observation figures
    timer = threading.Timer(timeout_s, expire)
    timer.daemon = True
    timer.start()
    try:
        res = rollout_sample(rank=rank, **kw)
    except Exception as ex:            # noqa: BLE001 - reported in the line, the observation figures stand
        res = {"error": repr(ex)}
    timer.cancel()
    return res


def dqn_sample(device, B=512, max_rails=100, depth=150):
    """The network's kernels on their own (N = 1 only): forward and one learn step at the script's batch size straight
    from uint8 observations, and act at batch 1; event-timed medians.  flop counts one multiply-add as two."""
    import types
    from flatland_marl_amd.agent import DQNAgent
    H3, T1 = 3 * max_rails, depth + 1
    a = types.SimpleNamespace(prediction_depth=depth, lr=5e-5, buffer_size=10, batch_size=B, update_every=10, gamma=0.99,
                              tau=1e-3)
    torch.manual_seed(0)
    agent = DQNAgent(a, action_space=2, bitmap_height=H3)
    g = torch.Generator().manual_seed(0)
    ps = (max_rails * T1 + 255) // 256 * 256
    # states and next states as the device replay memory hands them out: one buffer, the next states behind the states
    both = torch.randint(0, 3, (2 * B, 3, ps), generator=g, dtype=torch.uint8).to(device)
    obs, obs_n = both[:B], both[B:]
    act = torch.randint(0, 2, (B,), generator=g).to(device)
    r = torch.randn(B, generator=g).to(device)
    d = (torch.rand(B, generator=g) < 0.1).float().to(device)
    one = torch.randint(-1, 3, (1, 1, H3, T1), generator=g).float().to(device)

    def timed(fn, reps=10, warm=2):
        for _ in range(warm):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for x, y in ev:
            x.record()
            fn()
            y.record()
        torch.cuda.synchronize()
        return float(np.median([x.elapsed_time(y) for x, y in ev]))

    fwd_flop = (2.0 * H3 * T1 * 64 + 4.0 * 64 * H3 * 512) * B
    t_f = timed(lambda: agent.qnetwork_local.forward_obs(obs, ps))
    t_l = timed(lambda: agent.learn_obs((obs, act, r, obs_n, d), 0.99, ps))
    t_a = timed(lambda: agent.qnetwork_local(one), reps=30)
    # the same batch when the environments have 20 rails and the other 80 are padding (rows of zeros)
    both[:, :, 20 * T1:] = 0
    t_fr = timed(lambda: agent.qnetwork_local.forward_obs(obs, ps, rails_used=20))
    t_lr = timed(lambda: agent.learn_obs((obs, act, r, obs_n, d), 0.99, ps, rails_used=20))
    return {"workload": "batch %d, max_rails %d, depth %d, uint8 observations" % (B, max_rails, depth),
            "forward_ms": t_f, "forward_TFLOPs": fwd_flop / t_f / 1e9,
            "learn_step_ms": t_l, "learn_step_TFLOPs": 5 * fwd_flop / t_l / 1e9,
            "act_batch1_ms": t_a,
            "forward_20_of_%d_rails_ms" % max_rails: t_fr, "learn_step_20_of_%d_rails_ms" % max_rails: t_lr,
            "note": "fp32 values; the matrix products run on the bf16 matrix cores with every fp32 operand split into "
                    "three bf16 terms (six partial products: one sixth of the bf16 peak = 417 TFLOP/s bounds them; "
                    "fp32 matrix instructions peak at 157); a learn step is three forward passes, the backward pass "
                    "(about two more), Adam and the target's soft update"}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n_ranks, argv, timeout_s=1500.0):
    """`python bench.py --gpus N` without an outer launcher: start N fresh processes of this script, one rank per
    GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relay rank 0's output and return
    non-zero when any rank failed.  The parent never touches the GPU (no torch.cuda call, no library load), and no
    process is replaced by another: the ranks are children."""
    import subprocess
    import threading
    port = _free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FLM_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line.decode(errors="replace"))
            sys.stdout.flush()
    t = threading.Thread(target=relay, daemon=True)
    t.start()
    t0 = time.monotonic()
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [c for c in codes if c not in (None, 0)]
        if bad and failed is None:
            failed = time.monotonic()          # a rank died: the others may wait in a collective for ever
        if (failed is not None and time.monotonic() - failed > 20.0) or time.monotonic() - t0 > timeout_s:
            for p in procs:
                if p.poll() is None:
                    p.kill()                   # exactly the processes started above
        time.sleep(0.2)
    t.join(timeout=5.0)
    codes = [p.returncode for p in procs]
    if any(codes):
        sys.stderr.write("bench.py: rank exit codes %s\n" % codes)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=0, help="environments per GPU (default: the workload's)")
    ap.add_argument("--chunks", type=int, default=1, help="split the batch into this many sub-batches")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the sub-batches are spread over")
    ap.add_argument("--obs-serial", type=int, default=1,
                    help="1: observation kernels run one after the other on the main stream after the other stages; "
                         "2: pipelined, the other stages of the next sub-batch run on a side stream meanwhile")
    ap.add_argument("--paths", default="fused", choices=["fused", "dense"],
                    help="fused: paths and columns straight from the chain index (default when the tables fit the LDS); "
                         "dense: distance maps -> shortest paths -> columns as separate kernels")
    ap.add_argument("--no-store-paths", action="store_true", help="fused only: do not write the path elements to HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rollout", action="store_true", help="skip the env-steps/s sample of the batched rollout")
    ap.add_argument("--no-dqn", action="store_true", help="skip the timing of the network's kernels")
    ap.add_argument("--rollout-workload", default="cfg2g", choices=["cfg2", "cfg2g"])
    ap.add_argument("--rollout-max-rails", type=int, default=70)
    ap.add_argument("--rollout-envs", type=int, default=0, help="environments per GPU of the env-steps/s sample")
    ap.add_argument("--rollout-steps", type=int, default=40)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no outer launcher: this process only starts the ranks (before any GPU call) and waits for them
        return launch_ranks(args.gpus, sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()) if world > 1 else 0)
    torch.cuda.set_device(device)

    import flatland_marl_amd as fm
    from flatland_marl_amd import parallel
    from flatland_marl_amd.batch import plane_stride_for
    parallel.init_distributed("nccl", device)
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    n_ranks = torch.distributed.get_world_size() if dist_on else 1     # what the collectives really span
    backend = torch.distributed.get_backend() if dist_on else None
    if n_ranks != max(1, args.gpus) and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but the process group has %d ranks\n" % (args.gpus, n_ranks))

    data, E, A, depth, desc = workload_batch(args.workload, args.envs, seed=1000003 * rank)

    def measure(chunks, n_streams, obs_serial):
        """K timed passes of the observation path with the batch split into `chunks` sub-batches spread over
        `n_streams` HIP streams; returns the wall time, the sweep's average launch duration and the batch facts."""
        # The batch is processed as `chunks` sub-batches spread over `streams` HIP streams, so that the
        # latency-bound kernels of one sub-batch (graph, distances, paths) overlap the bandwidth-bound
        # observation kernel of another.
        n_chunks = max(1, min(chunks, E))
        streams = [torch.cuda.Stream(device=device) for _ in range(max(1, n_streams))] if n_streams > 1 or n_chunks > 1 \
            else [torch.cuda.current_stream(device)]
        main_stream = torch.cuda.current_stream(device)
        batches = []
        for ci in range(n_chunks):
            lo, hi = parallel.shard_range(E, ci, n_chunks)
            st = streams[ci % len(streams)]
            with torch.cuda.stream(st):
                b = fm.ObservationBatch(data["grid"][lo:hi], data["start"][lo:hi], data["direction"][lo:hi],
                                        data["target"][lo:hi], data["speed"][lo:hi], depth, device=device, stream=st)
                b.build_graph(sync=True)
            batches.append(b)
        Rp = max(b.Rp for b in batches)          # one rail padding for the whole batch
        for b in batches:
            b.max_rails = Rp
            b.Rp = Rp
            b.plane_stride = plane_stride_for(Rp, b.T1)
            with torch.cuda.stream(b.stream):
                b.reset_fused(sync=True)         # sizes the rail axis (one host read-back), allocates every buffer
        torch.cuda.synchronize()
        batch = batches[0]
        T1 = batch.T1

        ready = [torch.cuda.Event() for _ in batches]
        consumed = [None for _ in batches]

        fused = args.paths == "fused" and all(b.fused_fits() for b in batches)

        def front(b):
            """Everything before the observation sweep: switch graph, state numbering, chain index, then either the
            fused kernel (anchor distances + walks + columns in LDS) or dense distance maps -> walks -> columns."""
            b.build_graph(sync=False)
            if fused:
                b.chain_index()
                b.paths_columns_fused(shift=1, store_paths=not args.no_store_paths)
            else:
                b.compute_distances()
                b.shortest_paths()
                b.columns(shift=1)

        def step_pipelined(ev=None):
            # side stream: graph .. columns of sub-batch i+1 while the main stream writes the observations of sub-batch i
            side = streams[0]
            for ci, b in enumerate(batches):
                keep, b.stream = b.stream, side
                with torch.cuda.stream(side):
                    if consumed[ci] is not None:
                        side.wait_event(consumed[ci])       # the previous step's observation kernel has read the columns
                    front(b)
                    ready[ci].record(side)
                main_stream.wait_event(ready[ci])
                b.stream = main_stream
                if ev is not None:
                    ev[ci][0].record(main_stream)
                b.observations_fused()
                if ev is not None:
                    ev[ci][1].record(main_stream)
                consumed[ci] = torch.cuda.Event()
                consumed[ci].record(main_stream)
                b.stream = keep

        def step(ev=None):
            if obs_serial == 2:
                return step_pipelined(ev)
            for st in streams:
                st.wait_stream(main_stream)
            for ci, b in enumerate(batches):
                with torch.cuda.stream(b.stream):
                    front(b)
                    if not obs_serial:
                        if ev is not None:
                            ev[ci][0].record(b.stream)
                        b.observations_fused()
                        if ev is not None:
                            ev[ci][1].record(b.stream)
            for st in streams:
                main_stream.wait_stream(st)
            if obs_serial:
                for ci, b in enumerate(batches):
                    if ev is not None:
                        ev[ci][0].record(main_stream)
                    keep, b.stream = b.stream, main_stream
                    b.observations_fused()
                    b.stream = keep
                    if ev is not None:
                        ev[ci][1].record(main_stream)

        for _ in range(args.warmup):
            step()
        events = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in batches]
                  for _ in range(args.steps)]

        def barrier():
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()

        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(events[k])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        barrier()
        dt = parallel.max_over_ranks(dt, device)
        for b in batches:
            b.check_graph()
        n_err = int(sum((b.agent_err != 0).sum().item() for b in batches))
        # the observation kernel is launched once per sub-batch: average duration of one launch
        obs_ms = float(np.mean([a.elapsed_time(b) for per_step in events for a, b in per_step]))
        E_launch = batches[0].E

        res = dict(dt=dt, obs_ms=obs_ms, E_launch=batches[0].E, Rp=int(Rp), T1=T1, n_err=n_err, n_chunks=n_chunks,
                   n_streams=len(streams), fused=fused, plane_stride=batch.plane_stride)
        return res

    m = measure(args.chunks, args.streams, args.obs_serial)
    torch.cuda.empty_cache()
    dt, obs_ms, E_launch, Rp, T1, n_err = m["dt"], m["obs_ms"], m["E_launch"], m["Rp"], m["T1"], m["n_err"]
    n_chunks, fused = m["n_chunks"], m["fused"]
    out = {}
    if rank == 0:
        total_obs = E * A * world * args.steps
        alg_bytes = 4.0 * Rp * T1 * E_launch * A   # per launch of the observation kernel
        achieved = alg_bytes / (obs_ms * 1e-3) / 1e9
        # HBM bytes one launch really moves, from the PMC counters of a profiled run of this workload (the file
        # names the rocprofv3 summary it was taken from and the launch shape it was measured on); the kernel
        # writes 3 planes per agent and reads only the column form, so the figure scales with the bytes written
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "obs_kernel_traffic_%s.json" % args.workload)
        if os.path.exists(tfile):
            tj = json.load(open(tfile))
            traffic = tj.get("hbm_bytes_per_launch")
            if traffic is not None:
                measured_on = float(tj.get("E", 4096)) * tj.get("A", A) * tj.get("plane_stride", m["plane_stride"])
                traffic = traffic * (float(E_launch) * A * m["plane_stride"]) / measured_on
        physical = None if traffic is None else traffic / (obs_ms * 1e-3) / 1e9
        out = {
            "metric": "agent-observations/sec (bitmap+heatmaps)",
            "value": total_obs / dt,
            "unit": "agent-observations/s",
            "n_gpus": n_ranks,
            "dist_backend": backend,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i8",
            "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, desc), "envs_per_gpu": E, "agents": A,
                       "rails_padded": int(Rp), "depth": depth, "agents_flagged": n_err, "sub_batches": n_chunks, "streams": m["n_streams"],
                       "paths": ("fused" if fused else "dense") + ("" if not (fused and args.no_store_paths) else ", path elements not stored"),
                       "step": "graph build + chain index + anchor distances + shortest paths + occupancy columns + observations (dense planes are plane 0 of each observation)"},
            "roofline": {"kernel": "observations_columns_kernel", "bound": "hbm", "achieved": achieved, "peak": 8000.0,
                         "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                         "achieved_physical": physical, "frac_physical": None if physical is None else physical / 8000.0,
                         "note": "achieved/frac price the launch at the algorithmic 4*Rp*T1 bytes per agent-observation "
                                 "(one plane read + three written); the kernel never reads a dense plane, so it moves "
                                 "only `traffic` bytes (PMC): achieved_physical/frac_physical are that figure over the "
                                 "same event-timed duration",
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": obs_ms},
            "observations_only": {"value": E_launch * A / (obs_ms * 1e-3), "unit": "agent-observations/s",
                                  "note": "heat/observation sweep alone, one GPU"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(data, depth, int(Rp))
    if not args.no_rollout:
        res = guarded_rollout_sample(out, rank, 300, device=device, workload=args.rollout_workload,
                                     max_rails=args.rollout_max_rails, world=world, envs=args.rollout_envs,
                                     steps=args.rollout_steps)
        out["env_steps"] = res
    if rank == 0 and world == 1 and not args.no_dqn:
        try:
            out["dqn"] = dqn_sample(device)
        except Exception as ex:                       # the observation figures stand without it
            out["dqn"] = {"error": repr(ex)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 1 if isinstance(out.get("env_steps"), dict) and "error" in out["env_steps"] else 0


if __name__ == "__main__":
    sys.exit(main())
