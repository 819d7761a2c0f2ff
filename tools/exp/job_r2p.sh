#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2p_tests.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r2p_rollout_prof" -o rollout -- python3 "$GRAFT_REPO_ROOT/tools/rollout_bench.py" --workload cfg2g --max-rails 70 --steps 40 --train > "$GRAFT_REPO_ROOT/gpurun_out/r2p_rollout_prof.log" 2>&1
