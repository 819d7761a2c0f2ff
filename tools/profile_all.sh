#!/bin/bash
# Collect rocprofv3 evidence on one GPU: kernel statistics of the bench step, the DQN kernels and
# the rollout, the MFMA-busy counters of the DQN kernels, and HBM traffic counters of the bench step.
# Every pass is its own run with its own time limit; the script stops at the first failure.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_c; rm -rf $O
mkdir -p $O
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/bench -o bench --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-rollout > $O/bench.log 2>&1
echo "bench stats done"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/bench_g -o bench --output-format csv -- python3 bench.py --workload cfg3g --steps 5 --warmup 1 --no-cpu-baseline --no-rollout > $O/bench_g.log 2>&1
echo "bench cfg3g stats done"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/dqn -o dqn --output-format csv -- python3 tools/dqn_bench.py > $O/dqn.log 2>&1
echo "dqn stats done"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/rollout -o rollout --output-format csv -- python3 tools/rollout_bench.py --workload cfg2 --max-rails 100 --steps 30 --train > $O/rollout.log 2>&1
echo "rollout stats done"
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/dqn_mfma -o dqn --output-format csv -- python3 tools/dqn_bench.py > $O/dqn_mfma.log 2>&1
echo "dqn mfma done"
timeout -s KILL 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/wr -o wr --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rollout > $O/wr.log 2>&1
echo "write pass done"
timeout -s KILL 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/rd -o rd --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rollout > $O/rd.log 2>&1
echo "read pass done"
find $O -name "*.csv" | head -40
