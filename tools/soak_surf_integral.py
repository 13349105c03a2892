"""Randomised comparison of the SURF integral image with the CPU checker: random geometries, channel counts,
batch sizes and contents for a number of seconds. Run it once per form:
    MIF_SURF_INTEGRAL=fused   python3 tools/soak_surf_integral.py --seconds 120
    MIF_SURF_INTEGRAL=twopass python3 tools/soak_surf_integral.py --seconds 60
(the switch is read when the library is first used; without it the library picks the form by batch size)."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    dev = torch.device("cuda:0")
    t0 = time.time()
    cases = bad = frames_checked = 0
    last = t0
    while time.time() - t0 < args.seconds:
        w = int(rng.integers(2, 900)) if rng.random() < 0.8 else 4 * int(rng.integers(1, 300))
        h = int(rng.integers(1, 600))
        c = int(rng.choice([1, 3]))
        nf = int(rng.integers(1, 48))
        kind = rng.random()
        if kind < 0.5:
            px = rng.integers(0, 256, (nf, h, w, c), dtype=np.uint8)
        elif kind < 0.8:
            px = np.full((nf, h, w, c), int(rng.integers(0, 256)), dtype=np.uint8)
            px[:, rng.integers(0, h), rng.integers(0, w)] = 255
        else:
            px = (rng.random((nf, h, w, c)) < 0.02).astype(np.uint8) * 255
        if c == 1:
            px = px[..., 0]
        ex = oda.SurfExtractor(w, h, c, batch=nf)
        d = torch.from_numpy(np.ascontiguousarray(px)).to(dev)
        S = orc.SurfOracle(w, h)
        for rep in range(2):                     # the handle's second launch as well
            ex.enqueue_batch(d.data_ptr(), nf)
            for f in {0, nf - 1, int(rng.integers(0, nf))}:
                S.run(px[f])
                if not np.array_equal(ex.integral(f).view(np.uint32), S.integral().view(np.uint32)):
                    bad += 1
                    print("MISMATCH w=%d h=%d c=%d nf=%d frame=%d rep=%d" % (w, h, c, nf, f, rep), flush=True)
                frames_checked += 1
        ex.close()
        cases += 1
        if time.time() - last > 20:
            last = time.time()
            print("%d cases, %d frames checked, %d bad, %d s" % (cases, frames_checked, bad, last - t0), flush=True)
    print("done: %d cases, %d frames checked, %d bad" % (cases, frames_checked, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
