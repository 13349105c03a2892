"""Kernel experiments on the batched SIFT extraction: a few rendered 1080p frames repeated to fill a batch,
pyramid + feature phases enqueued `reps` times. Run it under `rocprofv3 --kernel-trace --stats` (one run per
library variant, MIF_LIBRARY=...) and read the per-kernel averages with tools/kstats.py; by itself it prints
the stage times the handle measures with HIP events and a checksum of the results (to see at a glance that
a variant returns the same bits).

usage: exp_sift.py [--batch 256] [--reps 3] [--width 1920 --height 1080 --channels 3] [--distinct 4]"""
import argparse
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from synth import blob_image  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--channels", type=int, default=3)
ap.add_argument("--distinct", type=int, default=4)
ap.add_argument("--no-desc", action="store_true")
a = ap.parse_args()
frames = np.stack([blob_image(a.width, a.height, seed=1234, channels=a.channels, shift=(2 * t, t), wrap=True)
                   for t in range(a.distinct)])
import torch  # noqa: E402
import object_detection_amd as oda  # noqa: E402
dev = torch.device("cuda:0")
reps = (a.batch + a.distinct - 1) // a.distinct
px = torch.from_numpy(frames).to(dev).repeat((reps,) + (1,) * (frames.ndim - 1))[:a.batch].contiguous()
ex = oda.SiftExtractor(a.width, a.height, a.channels, batch=a.batch)
ex.enqueue(px.data_ptr(), a.batch, not a.no_desc)
ex.sync()
ex.set_timing(True)
acc = np.zeros((3, 4))
for _ in range(a.reps):
    ex.enqueue(px.data_ptr(), a.batch, not a.no_desc)
    acc += ex.stage_times()
acc /= a.reps
crc = 0
for f in range(min(a.distinct, a.batch)):
    rc, feat = ex.copy_results(f, want_descriptors=not a.no_desc)
    for arr in (feat.keypoints, feat.orientations, feat.descriptors):
        crc = zlib.crc32(np.ascontiguousarray(arr).tobytes(), crc)
print("lib %s batch %d: scale_space %.3f detector %.3f orient_desc %.3f ms  (per octave scale space %s)  crc %08x kp %s"
      % (os.environ.get("MIF_LIBRARY", "default"), a.batch, acc[:, 1].sum(), acc[:, 2].sum(), acc[:, 3].sum(),
         np.round(acc[:, 1], 3).tolist(), crc, feat.counts))
