#!/usr/bin/env python3
"""Prints the headline fields of bench JSON lines: show.py file.json ..."""
import json
import sys

for path in sys.argv[1:]:
    try:
        lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
        d = json.loads(lines[-1])
    except (OSError, IndexError, ValueError) as exc:
        print(path, "unreadable:", exc)
        continue
    out = [path, "%.1f %s" % (d.get("value", 0.0), d.get("unit", "")), "%.2f ms" % d.get("ms_per_step", 0.0)]
    if "stage_ms_per_batch" in d:
        out.append(" ".join("%s=%.2f" % (k, v) for k, v in d["stage_ms_per_batch"].items()))
    if d.get("value_host_to_host"):
        out.append("host %.1f" % d["value_host_to_host"]["value"])
    if d.get("matcher") and "value" in d["matcher"]:
        out.append("matcher %.0f / exact %.0f" % (d["matcher"]["value"], d["matcher"]["value_exact_fp32"]))
    if "true_neighbour_recovered" in d:
        out.append("hit %s" % d["true_neighbour_recovered"])
    if "verified" in d:
        out.append("verified %s" % d["verified"])
    print("  ".join(out))
