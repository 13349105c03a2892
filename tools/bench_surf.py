#!/usr/bin/env python3
"""SURF extraction on GPU 0: time per frame of the one-frame host call (image in, interest points +
64-d descriptors out) for the box / scene pair and a 1080p frame, and frames/s of batches of 1080p
BGR frames resident in HBM (mif_surf_enqueue_batch). With --stats the kernel durations of the batched
run are read from HIP events around the integral-image kernels.

python tools/bench_surf.py [--batch 64] [--steps 10]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--skip-single", action="store_true")
    args = ap.parse_args()
    from synth import blob_image
    frames = np.stack([blob_image(1920, 1080, seed=1234, channels=3, shift=(2 * t, t), wrap=True)
                       for t in range(min(args.batch, 8))])
    import torch
    import object_detection_amd as oda
    from oracle_lib import read_pnm
    out = {"single_frame_host_to_host_ms": {}}
    if not args.skip_single:
        cases = {"box 324x223": read_pnm(os.path.join(ROOT, "tests", "golden", "box.pgm")),
                 "scene 512x384": read_pnm(os.path.join(ROOT, "tests", "golden", "scene.pgm")),
                 "blobs 1920x1080 BGR": frames[0]}
        for name, img in cases.items():
            h, w = img.shape[:2]
            ex = oda.SurfExtractor(w, h, 1 if img.ndim == 2 else 3)
            for _ in range(3):
                pts, raw = ex.run(img)
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                pts, raw = ex.run(img)
            dt = (time.perf_counter() - t0) / n
            out["single_frame_host_to_host_ms"][name] = {"points": len(pts), "ms": dt * 1e3}
            ex.close()
    # batches resident in HBM
    dev = torch.device("cuda:0")
    B = args.batch
    px = torch.from_numpy(frames).to(dev)
    px = px.repeat((B + len(frames) - 1) // len(frames), 1, 1, 1)[:B].contiguous()
    ex = oda.SurfExtractor(1920, 1080, 3, batch=B)
    for _ in range(2):
        ex.enqueue_batch(px.data_ptr(), B)
    ex.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ex.enqueue_batch(px.data_ptr(), B)
    ex.sync()
    dt = (time.perf_counter() - t0) / args.steps
    pts, raw = ex.copy_points(0)
    out["batched"] = {"workload": "SURF extraction, %d frames 1920x1080 BGR per batch, resident in HBM" % B,
                      "frames_per_s": B / dt, "ms_per_batch": dt * 1e3, "points_frame0": len(pts)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
