#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_rollout.py tests/test_gpu_rollout_batched.py tests/test_gpu_eval_rollout.py tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q --timeout 500 --timeout-method thread > gpurun_out/r2s_tests.log 2>&1
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2s_bench.json 2> gpurun_out/r2s_bench.err
