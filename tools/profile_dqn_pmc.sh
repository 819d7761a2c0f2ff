#!/bin/bash
# One counter pass over tools/dqn_bench.py (arguments: tag, then up to 8 SQ counters); prints per-kernel averages.
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_$TAG
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc "$@" -d $O -o dqn --output-format csv -- python3 tools/dqn_bench.py > $O/dqn.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/dqn_counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0]
    if "gemm" in k or "conv_u8" in k:
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(k[:60], {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
