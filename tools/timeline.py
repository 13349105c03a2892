#!/usr/bin/env python3
"""Prints the timeline of a rocprofv3 --kernel-trace --memory-copy-trace run: long copies and, per
stream, the spans in which kernels ran back to back. usage: timeline.py <dir> <prefix> [min_copy_ms] [tail_ms]"""
import csv
import sys

d, pre = sys.argv[1], sys.argv[2]
min_copy = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
tail = float(sys.argv[4]) if len(sys.argv) > 4 else 400.0
cp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", ""))
      for r in csv.DictReader(open("%s/%s_memory_copy_trace.csv" % (d, pre)))]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], r["Stream_Id"])
      for r in csv.DictReader(open("%s/%s_kernel_trace.csv" % (d, pre)))]
end = max(e for _, e, *_ in ks)
t0 = end - int(tail * 1e6)
ev = []
for s, e, k in cp:
    if (e - s) / 1e6 >= min_copy and e > t0:
        ev.append((s, e, "COPY " + k))
# group kernels of one stream into phases by name family
def fam(n):
    for key in ("k_blur", "k_detect", "k_refine", "k_scan_rows", "k_emit", "k_orientation", "k_descriptor",
                "k_match_screen", "k_match_rank", "k_match_fin", "k_match_partial", "k_row_norms", "k_match_filter"):
        if key in n:
            return key
    return None
cur = {}
for s, e, n, q, st in sorted(ks):
    f = fam(n)
    if f is None or e < t0:
        continue
    key = (st, q)
    c = cur.get(key)
    if c and c[2] == f and s - c[1] < 50000:
        c[1] = e
    else:
        if c:
            ev.append((c[0], c[1], "s%s q%s %s" % (key[0], key[1], c[2])))
        cur[key] = [s, e, f]
for key, c in cur.items():
    ev.append((c[0], c[1], "s%s q%s %s" % (key[0], key[1], c[2])))
for s, e, what in sorted(ev):
    print("%9.2f %9.2f %8.2f  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, what))
