#!/usr/bin/env python3
"""SIFT extraction time per frame on GPU 0 through the host-buffer entry point (image in host memory,
keypoints + orientations + descriptors back in host memory), i.e. what SIFT::extract_features costs."""
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
    ex = oda.SiftExtractor(w, h, 1 if img.ndim == 2 else 3)
    for _ in range(3):
        rc, feats = ex.extract(img)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        rc, feats = ex.extract(img)
    dt = (time.perf_counter() - t0) / n
    print("%-22s %5d keypoints  %.3f ms per frame (host to host), device part %.3f ms" % (
        name, len(feats.keypoints), dt * 1e3, feats.elapsed_ms))
