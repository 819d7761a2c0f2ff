#!/bin/bash
# One GPU call: pipelined bench variants, profiles.
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-rollout --no-cpu-baseline --chunks 4 --obs-serial 2 --streams 2 > gpurun_out/r2g_bench_pipe4.json 2> gpurun_out/r2g_bench_pipe4.err
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-rollout --no-cpu-baseline --chunks 8 --obs-serial 2 --streams 2 > gpurun_out/r2g_bench_pipe8.json 2> gpurun_out/r2g_bench_pipe8.err
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-rollout --no-cpu-baseline --chunks 2 --obs-serial 0 --streams 2 > gpurun_out/r2g_bench_par2.json 2> gpurun_out/r2g_bench_par2.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2g_dqn_prof" -o dqn -- python3 "$GRAFT_REPO_ROOT/tools/dqn_bench.py" > "$GRAFT_REPO_ROOT/gpurun_out/r2g_dqn_prof.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2g_bench_prof" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 1 --no-rollout --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r2g_bench_prof.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2g_bench_prof_g" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload cfg3g --steps 5 --warmup 1 --no-rollout --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r2g_bench_prof_g.log" 2>&1
