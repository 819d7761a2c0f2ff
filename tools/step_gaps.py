"""Idle time of the device inside rollout steps, from a rocprofv3 kernel trace (development tool):
python tools/step_gaps.py <..._kernel_trace.csv> prints one late step kernel by kernel with the gap in front of each,
then the mean span and busy time of the last 30 steps (a step = from one env_step_kernel to the next)."""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "env_step_kernel" in r["Kernel_Name"]]
a, b = idx[-6], idx[-5]
prev = None
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-56s %7.1f us  gap %6.1f" % (r["Kernel_Name"][:56], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
    prev = e
span, busy = [], []
for k in range(len(idx) - 31, len(idx) - 1):
    a, b = idx[k], idx[k + 1]
    span.append((int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3)
    busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a + 1:b + 1]) / 1e3)
print("mean span %.1f us, busy %.1f us" % (np.mean(span), np.mean(busy)))
