#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_dqn.py tests/test_gpu_train.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2k_tests.log 2>&1
timeout -k 10 150 python tools/dqn_bench.py > gpurun_out/r2k_dqn_bench.log 2>&1
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-rollout --no-cpu-baseline --no-overlapped --chunks 4 --obs-serial 0 --streams 4 > gpurun_out/r2k_bench_par4s4_$rep.json 2> gpurun_out/r2k_err.log
done
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-rollout --no-cpu-baseline --no-overlapped --chunks 8 --obs-serial 0 --streams 4 > gpurun_out/r2k_bench_par8s4.json 2> gpurun_out/r2k_err.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-rollout --no-cpu-baseline --no-overlapped --chunks 2 --obs-serial 0 --streams 2 > gpurun_out/r2k_bench_par2s2.json 2> gpurun_out/r2k_err.log
timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-rollout --no-cpu-baseline > gpurun_out/r2k_bench_default.json 2> gpurun_out/r2k_err.log
