#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 700 python -u -m pytest tests/test_gpu_rollout.py tests/test_gpu_rollout_batched.py tests/test_gpu_fuzz.py tests/test_gpu_eval_rollout.py tests/test_gpu_cfg4.py tests/test_gpu_train.py -x -q --timeout 400 --timeout-method thread > gpurun_out/r2q_tests.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r2q_rollout_prof" -o rollout -- python3 "$GRAFT_REPO_ROOT/tools/rollout_bench.py" --workload cfg2g --max-rails 70 --steps 60 > "$GRAFT_REPO_ROOT/gpurun_out/r2q_rollout_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2q_bench.json 2> gpurun_out/r2q_bench.err
