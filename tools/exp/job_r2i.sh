#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_paths.py tests/test_gpu_fuzz.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2i_tests.log 2>&1
timeout -k 10 300 python -u tools/fused_phases.py 4096 cfg3 > gpurun_out/r2i_phases_cfg3.log 2>&1
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-rollout --no-cpu-baseline > gpurun_out/r2i_bench_cfg3.json 2> gpurun_out/r2i_bench_cfg3.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2i_dqn_prof" -o dqn -- python3 "$GRAFT_REPO_ROOT/tools/dqn_bench.py" > "$GRAFT_REPO_ROOT/gpurun_out/r2i_dqn_prof.log" 2>&1
