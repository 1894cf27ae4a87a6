#!/usr/bin/env python3
"""Average HBM traffic per launch from rocprofv3 --pmc passes.

  python tools/pmc_summary.py FETCH_DIR WRITE_DIR [out.json]

Each directory holds one pass (`rocprofv3 --pmc FETCH_SIZE ...` and `--pmc WRITE_SIZE ...`) of
`bench.py --pmc-target`.  FETCH_SIZE / WRITE_SIZE are kilobytes; on gfx950 FETCH_SIZE tallies a
128-byte request of a wide coalesced read as 64 bytes, so the read side is doubled.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"dsw::(?:\(anonymous namespace\)::)?(\w+)", name)
    if not m:
        return None
    args = re.search(r"<([^>]*)>", name)
    return m.group(1) + ("<" + args.group(1) + ">" if args else "")


def per_kernel(d, counter):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + d)
    acc = defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != counter:
            continue
        k = short(row["Kernel_Name"])
        if k:
            acc[k].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch, n = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(fetch):
        rd = fetch[k] * 1024 * 2
        wr = write.get(k, 0.0) * 1024
        out[k] = dict(launches=n[k], read_bytes=rd, write_bytes=wr, traffic=rd + wr)
        print("%-70s n=%3d  read %9.2f MB  write %9.2f MB" % (k[:70], n[k], rd / 1e6, wr / 1e6))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
