"""Generates tests/golden/sift_golden.npz from the CPU checker in oracle/.

The reference ships no golden vectors for SIFT and cannot run in this
environment, so these tables pin the checker against regressions (and give the
GPU tests a file to compare with when the checker library is unavailable); they
are NOT outputs of the reference. Run: python tests/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_lib as orc  # noqa: E402
from synth import blob_image, dense_image  # noqa: E402

SURF_CASES = ["box", "scene", "blob_bgr_200x150"]
CASES = ["box", "scene", "blob_256x192", "blob_bgr_200x150", "dense_256x192", "blob_bgr_97x61"]


def load_case(name):
    if name in ("box", "scene"):
        return orc.read_pnm(os.path.join(orc.GOLDEN, name + ".pgm"))
    if name == "blob_256x192":
        return blob_image(256, 192, seed=1)
    if name == "blob_bgr_200x150":
        return blob_image(200, 150, seed=2, channels=3)
    if name == "dense_256x192":
        return dense_image(256, 192, seed=0)
    if name == "blob_bgr_97x61":
        return blob_image(97, 61, seed=22, channels=3)
    raise KeyError(name)


def main():
    out = {}
    for name in CASES:
        img = load_case(name)
        h, w = img.shape[:2]
        S = orc.SiftOracle(w, h, 1 if img.ndim == 2 else 3)
        rc = S.run(img)
        kp, octs, ori, des = S.all_features()
        out[name + "_rc"] = np.int32(rc)
        out[name + "_kp"], out[name + "_oct"], out[name + "_ori"], out[name + "_des"] = kp, octs, ori, des
        for o in range(3):
            out["%s_dogsum%d" % (name, o)] = S.dog(o).view(np.uint32).sum(dtype=np.uint64)
        print(name, img.shape, "rc", rc, "keypoints", [S.count(o) for o in range(3)],
              "raw", [S.raw_count(o) for o in range(3)])
    np.savez_compressed(os.path.join(orc.GOLDEN, "sift_golden.npz"), **out)
    out = {}
    for name in SURF_CASES:
        img = load_case(name)
        h, w = img.shape[:2]
        S = orc.SurfOracle(w, h)
        n = S.run(img)
        out[name] = S.points()
        out[name + "_intsum"] = S.integral().view(np.uint32).sum(dtype=np.uint64)
        print("surf", name, img.shape, "points", n)
    np.savez_compressed(os.path.join(orc.GOLDEN, "surf_golden.npz"), **out)


if __name__ == "__main__":
    main()
