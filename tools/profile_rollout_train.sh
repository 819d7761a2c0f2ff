#!/bin/bash
# Kernel statistics of rollout steps with learning (the bench's 1024 x 10 workload, at most 4 learn steps per rollout
# step) and the host profile beside it: what the transition-store path launches.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_train
rm -rf $O; mkdir -p $O
timeout -s KILL 300 python3 tools/rollout_hostprof.py --steps 60 --train > $O/host.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O -o train --output-format csv -- python3 tools/rollout_hostprof.py --steps 60 --train > $O/train.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/train_kernel_stats.csv", recursive=True)[0]
tot = 0
rows = list(csv.DictReader(open(f)))
for row in rows:
    tot += float(row["TotalDurationNs"])
print("total kernel time per step (65 steps incl. warmup): %.1f us" % (tot / 65 / 1e3))
for row in rows[:40]:
    print("%-60s %5s calls %8.1f us avg %8.1f us/step" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 65 / 1e3))
PY
head -34 $O/host.log
