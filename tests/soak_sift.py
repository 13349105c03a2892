#!/usr/bin/env python3
"""Randomised parity soak: the HIP SIFT path against the CPU checker (oracle/) on images of random sizes,
channel counts and contents, single frames and small batches, with and without up_scale; every output must be
bit-identical. Test infrastructure like the rest of this directory (it uses the checker), but not collected by
pytest: it runs for as long as --seconds says; a failure prints the case.
usage: python tests/soak_sift.py --seconds 120 [--seed 1] [--wide]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import object_detection_amd as oda  # noqa: E402
import oracle_lib as orc  # noqa: E402
from synth import blob_image, dense_image  # noqa: E402


def same(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--wide", action="store_true",
                    help="images 1000 .. 4096 pixels wide and 40 .. 160 high: large column offsets at a small cost")
    args = ap.parse_args()
    import torch
    rng = np.random.RandomState(args.seed)
    t0, cases, keypoints, bad = time.time(), 0, 0, 0
    while time.time() - t0 < args.seconds:
        w, h = int(rng.randint(24, 420)), int(rng.randint(24, 320))
        if args.wide:
            w, h = int(rng.randint(1000, 4097)), int(rng.randint(40, 161))
        c = int(rng.choice([1, 3]))
        up = bool(rng.rand() < 0.15) and w * h < 60000
        batch = int(rng.choice([1, 1, 2, 3]))
        kind = rng.choice(["blob", "dense", "big"])
        if kind == "dense":
            c = 1                    # (the dense pattern is a gray image)
        frames = []
        for b in range(batch):
            seed = int(rng.randint(1 << 30))
            if kind == "dense":
                img = dense_image(w, h, seed=seed)
            elif kind == "big":      # few large blobs: keypoints at coarse scales, wide descriptor windows
                img = blob_image(w, h, seed=seed, n_blobs=max(3, w * h // 6000), channels=c)
            else:
                img = blob_image(w, h, seed=seed, channels=c, shift=(b, 2 * b))
            frames.append(img)
        what = "w %d h %d c %d up %s batch %d %s" % (w, h, c, up, batch, kind)
        try:
            if batch == 1:
                ex = oda.SiftExtractor(w, h, c, up_scale=up)
                got = [ex.extract(frames[0])]
            else:
                if up:
                    continue
                px = torch.from_numpy(np.stack(frames)).to(torch.device("cuda:0"))
                ex = oda.SiftExtractor(w, h, c, batch=batch)
                ex.enqueue(px.data_ptr(), batch)
                got = [ex.copy_results(f) for f in range(batch)]
            for f, img in enumerate(frames):
                S = orc.SiftOracle(w, h, c, up_scale=up)
                rc_o = S.run(img)
                rc_g, feat = got[f]
                ok = rc_g == rc_o
                if ok and rc_o == 0:
                    kp, octs, ori, des = S.all_features()
                    ok = (feat.counts == [S.count(o) for o in range(3)] and same(feat.keypoints, kp) and
                          same(feat.orientations, ori) and same(feat.descriptors, des))
                    keypoints += len(kp)
                if not ok:
                    bad += 1
                    print("MISMATCH:", what, "frame", f, flush=True)
            cases += 1
        except Exception as exc:        # a geometry one side rejects: both must
            print("case failed:", what, repr(exc), flush=True)
            bad += 1
        if cases % 50 == 0:
            print("%d cases, %d keypoints, %d bad, %.0f s" % (cases, keypoints, bad, time.time() - t0), flush=True)
    print("done: %d cases, %d keypoints, %d bad" % (cases, keypoints, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
