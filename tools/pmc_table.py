#!/usr/bin/env python3
"""Per-kernel averages of every counter in rocprofv3 --pmc output directories.

  python tools/pmc_table.py DIR [DIR ...]
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                m = re.search(r"dsw::(?:\(anonymous namespace\)::)?(\w+)", row["Kernel_Name"])
                if not m:
                    continue
                a = re.search(r"<([^>]*)>", row["Kernel_Name"])
                k = m.group(1) + ("<" + a.group(1) + ">" if a else "")
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("    %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
