#!/usr/bin/env python3
"""Summary of a rocprofv3 kernel trace (the *_kernel_trace.csv of --kernel-trace): time and launches per kernel name of
the LAST of n equal steps (the launches are split into n equal runs), names cut to their first 90 characters.
usage: tools/trace_summary.py TRACE.csv [n_steps]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adam_update_kernel" in r["Kernel_Name"]]
if marks:   # a training loop: a step ends with the optimizer's launches
    ends = [i for j, i in enumerate(marks) if j + 1 == len(marks) or marks[j + 1] - i > 50]
    for a, b in zip([-1] + ends[:-1], ends):
        seg = rows[a + 1:b + 1]
        print("step: %d launches, busy %.2f ms" % (len(seg), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6))
    last = rows[ends[-2] + 1:ends[-1] + 1]
else:
    per = len(rows) // n
    last = rows[len(rows) - per:]
t = defaultdict(lambda: [0, 0])
for r in last:
    k = r["Kernel_Name"][:90]
    t[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    t[k][1] += 1
tot = sum(v[0] for v in t.values())
span = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
print("last step: %d launches, busy %.2f ms, span %.2f ms" % (len(last), tot / 1e6, span / 1e6))
for k, v in sorted(t.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print("%8.3f ms %5.1f%% %5d  %s" % (v[0] / 1e6, 100.0 * v[0] / tot, v[1], k))
