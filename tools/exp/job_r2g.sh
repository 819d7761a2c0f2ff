#!/bin/bash
# One GPU call: parity of the chain index change, the bounds-checked rollout run, cfg3 / cfg3g bench lines, DQN profile.
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_paths.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r2g_parity.log 2>&1
timeout -k 10 300 python -u tools/exp/run_checked_rows.py > gpurun_out/r2_checked_rows.log 2>&1
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-rollout --no-cpu-baseline > gpurun_out/r2g_bench_cfg3.json 2> gpurun_out/r2g_bench_cfg3.err
timeout -k 10 300 python bench.py --workload cfg3g --steps 5 --warmup 1 --no-rollout --no-cpu-baseline > gpurun_out/r2g_bench_cfg3g.json 2> gpurun_out/r2g_bench_cfg3g.err
for cus in 64 96 128; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-rollout --no-cpu-baseline --chunks 4 --obs-serial 2 --streams 2 --front-cus $cus > gpurun_out/r2g_bench_cu${cus}.json 2> gpurun_out/r2g_bench_cu${cus}.err
done
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-rollout --no-cpu-baseline --chunks 4 --obs-serial 2 --streams 2 > gpurun_out/r2g_bench_pipe4.json 2> gpurun_out/r2g_bench_pipe4.err
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-rollout --no-cpu-baseline --chunks 4 --obs-serial 2 --streams 2 --front-cus 96 --cu-pattern low > gpurun_out/r2g_bench_cu96low.json 2> gpurun_out/r2g_bench_cu96low.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2g_dqn_prof" -o dqn -- python3 "$GRAFT_REPO_ROOT/tools/dqn_bench.py" > "$GRAFT_REPO_ROOT/gpurun_out/r2g_dqn_prof.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2g_bench_prof" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 1 --no-rollout --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r2g_bench_prof.log" 2>&1
