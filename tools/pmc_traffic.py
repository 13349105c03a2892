#!/usr/bin/env python3
"""Merges a FETCH_SIZE pass and a WRITE_SIZE pass of `rocprofv3 --pmc` over bench.py into per-launch
memory-side bytes per kernel, and the scale-space (k_blur_level + k_blur_pair) total per frame.

usage: pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv frames_per_batch out.json

FETCH_SIZE / WRITE_SIZE are reported in units of 1024 bytes by rocprofv3. On gfx950 FETCH_SIZE tallies
the 128-byte requests of coalesced reads at 64 bytes, i.e. reports half of the bytes (checked here on
kernels with known byte counts: k_gray_bgr_x4 reads 3 B/px, k_decimate touches every other row; both
come out at 0.50 of their bytes), so it is doubled; WRITE_SIZE is exact for these stores. The JSON keeps
the raw fetch figure next to the corrected one.
"""
import collections
import csv
import json
import sys


FETCH_FACTOR = 2.0        # gfx950: FETCH_SIZE counts a 128-byte request as 64 bytes


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        tot[name] += float(r["Counter_Value"]) * 1024.0
        calls[name].add(r["Dispatch_Id"])
    return {k: (v, len(calls[k])) for k, v in tot.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    frames = int(sys.argv[3])
    out = {"frames_per_batch": frames, "kernels": {}}
    blur_bytes = blur_launches = last_level_launches = 0.0
    for k in sorted(set(fetch) | set(write)):
        fb, fc = fetch.get(k, (0.0, 0))
        wb, wc = write.get(k, (0.0, 0))
        calls = max(fc, wc)
        out["kernels"][k] = {"launches": calls, "fetch_bytes_per_launch": FETCH_FACTOR * fb / max(fc, 1),
                             "fetch_size_raw_per_launch": fb / max(fc, 1),
                             "write_bytes_per_launch": wb / max(wc, 1)}
        if k.startswith("k_blur_level") or k.startswith("k_blur_pair"):
            blur_bytes += FETCH_FACTOR * fb / max(fc, 1) * calls + wb / max(wc, 1) * calls
            blur_launches += calls
            if k.startswith("k_blur_level<11"):          # the last level: one launch per octave and batch
                last_level_launches = calls
    batches = last_level_launches / 3.0
    out["blur_bytes_per_frame"] = blur_bytes / batches / frames
    out["blur_launches_per_batch"] = blur_launches / batches
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print("blur bytes per frame: %.1f MB over %g launches per batch"
          % (out["blur_bytes_per_frame"] / 1e6, out["blur_launches_per_batch"]))


if __name__ == "__main__":
    main()
