#!/bin/bash
# One GPU call: parity after the ranking / register changes, DQN tests, phases, bench lines.
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_paths.py tests/test_gpu_dqn.py tests/test_gpu_train.py tests/test_gpu_fuzz.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2h_tests.log 2>&1
timeout -k 10 150 python tools/dqn_bench.py > gpurun_out/r2h_dqn_bench.log 2>&1
timeout -k 10 300 python -u tools/fused_phases.py 4096 cfg3 > gpurun_out/r2h_phases_cfg3.log 2>&1
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-rollout --no-cpu-baseline > gpurun_out/r2h_bench_cfg3.json 2> gpurun_out/r2h_bench_cfg3.err
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-rollout --no-cpu-baseline --chunks 2 --obs-serial 0 --streams 2 > gpurun_out/r2h_bench_par2.json 2> gpurun_out/r2h_bench_par2.err
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-rollout --no-cpu-baseline --chunks 4 --obs-serial 0 --streams 2 > gpurun_out/r2h_bench_par4.json 2> gpurun_out/r2h_bench_par4.err
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-rollout --no-cpu-baseline --chunks 4 --obs-serial 0 --streams 4 > gpurun_out/r2h_bench_par4s4.json 2> gpurun_out/r2h_bench_par4s4.err
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-rollout --no-cpu-baseline --chunks 3 --obs-serial 0 --streams 3 > gpurun_out/r2h_bench_par3s3.json 2> gpurun_out/r2h_bench_par3s3.err
