#!/usr/bin/env python3
"""SURF extraction time per frame on GPU 0 (host image in, interest points + 64-d descriptors out)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import object_detection_amd as oda  # noqa: E402
from oracle_lib import read_pnm  # noqa: E402
from synth import blob_image  # noqa: E402

cases = {"box 324x223": read_pnm(os.path.join(ROOT, "tests", "golden", "box.pgm")),
         "scene 512x384": read_pnm(os.path.join(ROOT, "tests", "golden", "scene.pgm")),
         "blobs 1920x1080 BGR": blob_image(1920, 1080, seed=1234, channels=3)}
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
    print("%-22s %5d points  %.3f ms per frame (host to host)" % (name, len(pts), dt * 1e3))
