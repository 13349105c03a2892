#!/usr/bin/env python3
"""Memory-side bytes of the SURF integral-image launches from a FETCH_SIZE pass and a WRITE_SIZE pass of
`rocprofv3 --pmc` over tools/bench_surf.py (tools/profile_round.sh, step `surf`).

usage: surf_traffic.py fetch_counter_collection.csv write_counter_collection.csv frames_per_batch out.json

Units and the gfx950 correction of FETCH_SIZE as in tools/pmc_traffic.py."""
import json
import sys

from pmc_traffic import FETCH_FACTOR, per_kernel

KERNELS = ("k_integral_fused", "k_scan_rows", "k_scan_cols")


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    frames = int(sys.argv[3])
    out = {"source": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/bench_surf.py --batch %d; "
                     "FETCH_SIZE doubled, the gfx950 correction" % frames,
           "frames": frames, "fetch_bytes": {}, "write_bytes": {}}
    total = 0.0
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith(KERNELS):
            continue
        fb, fc = fetch.get(k, (0.0, 0))
        wb, wc = write.get(k, (0.0, 0))
        out["fetch_bytes"][k] = FETCH_FACTOR * fb / max(fc, 1)
        out["write_bytes"][k] = wb / max(wc, 1)
        total += out["fetch_bytes"][k] + out["write_bytes"][k]
    out["integral_bytes_per_frame"] = total / frames
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print("integral image: %.2f MB per frame (%s)" % (total / frames / 1e6, ", ".join(out["fetch_bytes"])))


if __name__ == "__main__":
    main()
