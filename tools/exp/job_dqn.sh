#!/bin/bash
# DQN kernels: tests, timings, kernel statistics.
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_dqn.py tests/test_gpu_train.py tests/test_gpu_trainer_ranks.py -x -q --timeout 300 --timeout-method thread > gpurun_out/dqn_tests.log 2>&1
timeout -k 10 150 python tools/dqn_bench.py > gpurun_out/dqn_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/dqn_prof" -o dqn -- python3 "$GRAFT_REPO_ROOT/tools/dqn_bench.py" > "$GRAFT_REPO_ROOT/gpurun_out/dqn_prof.log" 2>&1
