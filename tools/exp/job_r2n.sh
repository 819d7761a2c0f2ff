#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_paths.py tests/test_gpu_fuzz.py tests/test_gpu_docs_kat.py tests/test_gpu_rollout.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2n_tests.log 2>&1
timeout -k 10 300 python -u tools/fused_phases.py 4096 cfg3 > gpurun_out/r2n_phases_cfg3.log 2>&1
timeout -k 10 300 python -u tools/fused_phases.py 4096 cfg3g > gpurun_out/r2n_phases_cfg3g.log 2>&1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-rollout --no-cpu-baseline > gpurun_out/r2n_bench_cfg3.json 2> gpurun_out/r2n_bench_cfg3.err
timeout -k 10 300 python bench.py --workload cfg3g --steps 10 --warmup 2 --no-rollout --no-cpu-baseline > gpurun_out/r2n_bench_cfg3g.json 2> gpurun_out/r2n_bench_cfg3g.err
timeout -k 10 600 python -u -m pytest tests/test_gpu_fullsize.py -x -q --timeout 500 --timeout-method thread > gpurun_out/r2n_fullsize.log 2>&1
