#!/usr/bin/env python3
"""Pretty-print the JSON line of bench.py (stdin): step time and the per-kernel table."""
import json
import sys

for line in sys.stdin:
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print("value %.1f %s   ms_per_step %.3f   n_gpus %d" % (d["value"], d["unit"], d["ms_per_step"], d["n_gpus"]))
    for k, v in d.get("kernels", {}).items():
        if "achieved" in v:
            print("  %-18s %9.1f us  %9.1f %-8s  frac %.3f" % (k, v["us"], v["achieved"], v["unit"], v["frac"]))
        else:
            print("  %-18s %9.1f us" % (k, v["us"]))
    if "cpu_baseline" in d:
        print("  cpu_baseline", d["cpu_baseline"])
