#!/bin/bash
# Whole GPU suite, the default bench line, then the profile passes.
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1500 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/full_tests.log 2>&1
timeout -k 10 600 python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err
timeout -k 10 300 python bench.py --workload cfg3g --no-rollout > gpurun_out/full_bench_cfg3g.json 2> gpurun_out/full_bench_cfg3g.err
bash tools/profile_all.sh > gpurun_out/full_profile.log 2>&1
