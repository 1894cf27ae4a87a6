#!/usr/bin/env python3
"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py.

usage: step_breakdown.py <kernel_trace.csv> [step_index]
A step is delimited by consecutive launches of the window-selection kernel (one per DS-MSA
forward); prints every kernel of that step with its launch count and summed duration.  Launches are keyed by
kernel AND grid size, so that the shapes one kernel runs at in a step (e.g. the attention at the WindowedMSA
shape, 3072 workgroups, and at the DS-MSA shape, 768) are separate lines."""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "select_rank_kernel" in r["Kernel_Name"]]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 2
    seg = rows[marks[k]:marks[k + 1]]
    wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    agg = collections.OrderedDict()
    for r in seg:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        grid = "x".join(str(int(r[c]) // max(1, int(r.get(c.replace("Grid", "Workgroup"), 1) or 1)))
                        for c in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if c in r and int(r[c] or 1) > 1) or "1"
        a = agg.setdefault((r["Kernel_Name"][:92], grid), [0, 0.0])
        a[0] += 1
        a[1] += d
    busy = sum(v[1] for v in agg.values())
    print("step %d: %d launches, wall %.1f us, GPU busy %.1f us" % (k, len(seg), wall, busy))
    print("%-92s %12s %6s %10s" % ("kernel", "workgroups", "calls", "sum_us"))
    for (n, grid), (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-92s %12s %6d %10.1f" % (n, grid, c, d))


if __name__ == "__main__":
    main()
