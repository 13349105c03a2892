"""Runs the 1080p SIFT extraction a few times (for rocprofv3 --kernel-trace --stats)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import object_detection_amd as oda  # noqa: E402
from synth import blob_image  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
img = blob_image(w, h, seed=1234, channels=3)
ex = oda.SiftExtractor(w, h, 3)
for i in range(3):
    ex.extract(img)
t = time.time()
ms = []
for i in range(n):
    rc, f = ex.extract(img)
    ms.append(f.elapsed_ms)
dt = time.time() - t
print("frames", n, "wall ms/frame %.3f" % (dt / n * 1e3), "gpu ms/frame median %.3f min %.3f" % (np.median(ms), min(ms)),
      "keypoints", f.counts)
