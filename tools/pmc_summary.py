#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters: pmc_summary.py counter_collection.csv [name-filter]."""
import collections
import csv
import sys


def main():
    rows = csv.DictReader(open(sys.argv[1]))
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if flt and flt not in name:
            continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name].add(r["Dispatch_Id"])
    for name, c in sorted(acc.items()):
        n = len(calls[name])
        print("%-34s calls %3d  " % (name[:34], n) + "  ".join("%s=%.4g" % (k, v / n) for k, v in sorted(c.items())))


if __name__ == "__main__":
    main()
