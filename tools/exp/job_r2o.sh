#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -u tools/rollout_hostprof.py --steps 60 > gpurun_out/r2o_hostprof.log 2>&1
timeout -k 10 300 python -u tools/rollout_bench.py --workload cfg2g --max-rails 70 --steps 60 > gpurun_out/r2o_rollout_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r2o_rollout_prof" -o rollout -- python3 "$GRAFT_REPO_ROOT/tools/rollout_bench.py" --workload cfg2g --max-rails 70 --steps 60 > "$GRAFT_REPO_ROOT/gpurun_out/r2o_rollout_prof.log" 2>&1
