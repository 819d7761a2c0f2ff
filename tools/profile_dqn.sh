#!/bin/bash
# Kernel statistics of tools/dqn_bench.py under rocprofv3 (arguments: output tag, then dqn_bench.py's own); prints the
# top kernels.
set -e
TAG=${1:-dqn}
shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$TAG
mkdir -p $O
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O -o dqn --output-format csv -- python3 tools/dqn_bench.py "$@" > $O/dqn.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/dqn_kernel_stats.csv", recursive=True)[0]
for i, row in enumerate(csv.DictReader(open(f))):
    if i >= 14: break
    print("%-70s %5s calls %9.1f us avg %5s%%" % (row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e3, row["Percentage"]))
PY
