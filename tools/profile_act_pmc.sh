#!/bin/bash
# One counter pass over act-only rollout steps (arguments: tag, kernel-name filter, then up to 8 SQ counters).
set -e
TAG=$1; FILT=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_act_$TAG
rm -rf $O; mkdir -p $O
timeout -s KILL 200 rocprofv3 --pmc "$@" -d $O -o act --output-format csv -- python3 tools/rollout_hostprof.py --steps 20 > $O/act.log 2>&1
python3 - "$O" "$FILT" <<'PY'
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/act_counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0]
    if sys.argv[2] in k:
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(k[:60], {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
