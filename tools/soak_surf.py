"""Randomised comparison of the whole SURF pipeline with the CPU checker: random geometries, channel counts,
batch sizes and contents (blobs, noise, axis-aligned rectangles on a flat ground: many Haar responses that
are exactly zero) for a number of seconds. Every field of every interest point of the probed frames must
equal the checker's bits.
    python3 tools/soak_surf.py --seconds 120 [--seed N]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import object_detection_amd as oda  # noqa: E402
import oracle_lib as orc  # noqa: E402
from synth import blob_image  # noqa: E402


def render(rng, w, h, c):
    kind = rng.random()
    if kind < 0.45:
        return blob_image(w, h, seed=int(rng.integers(1, 1 << 30)), channels=c)
    if kind < 0.6:
        img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    else:
        img = np.full((h, w, c), int(rng.integers(0, 200)), dtype=np.uint8)
        for _ in range(int(rng.integers(1, 30))):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            x1, y1 = min(w, x0 + int(rng.integers(2, 60))), min(h, y0 + int(rng.integers(2, 60)))
            img[y0:y1, x0:x1] = rng.integers(0, 256, c)
    return img if c == 3 else img[..., 0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    dev = torch.device("cuda:0")
    t0 = last = time.time()
    cases = bad = points = 0
    while time.time() - t0 < args.seconds:
        w, h = int(rng.integers(40, 500)), int(rng.integers(40, 400))
        c = int(rng.choice([1, 3]))
        nf = int(rng.choice([1, 2, 5, 40, 70]))
        distinct = [render(rng, w, h, c) for _ in range(min(nf, 3))]
        frames = np.stack([distinct[i % len(distinct)] for i in range(nf)])
        ex = oda.SurfExtractor(w, h, c, batch=nf)
        ex.enqueue_batch(torch.from_numpy(frames).to(dev).data_ptr(), nf)
        S = orc.SurfOracle(w, h)
        for f in sorted({0, nf - 1, min(nf - 1, 2)}):
            n = S.run(frames[f])
            pts, raw = ex.copy_points(f)
            ok = len(pts) == n and raw == S.raw_count()
            if ok and n:
                P = S.points()
                got = np.concatenate([pts["x"][:, None], pts["y"][:, None], pts["scale"][:, None],
                                      pts["orientation"][:, None]], axis=1)
                ok = (np.array_equal(got.view(np.uint32), P[:, :4].view(np.uint32)) and
                      np.array_equal(pts["laplacian"], P[:, 4].astype(np.int32)) and
                      np.array_equal(pts["descriptor"].view(np.uint32), P[:, 6:].view(np.uint32)))
            if not ok:
                bad += 1
                print("MISMATCH w=%d h=%d c=%d nf=%d frame=%d" % (w, h, c, nf, f), flush=True)
            points += n
        ex.close()
        cases += 1
        if time.time() - last > 20:
            last = time.time()
            print("%d cases, %d points, %d bad, %d s" % (cases, points, bad, last - t0), flush=True)
    print("done: %d cases, %d points, %d bad" % (cases, points, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
