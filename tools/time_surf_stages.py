"""Stage times (HIP events at the stage boundaries) of a SURF batch of rendered 1080p BGR frames.
usage: python3 tools/time_surf_stages.py [batch]"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import object_detection_amd as oda  # noqa: E402
from synth import blob_image  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = np.stack([blob_image(1920, 1080, seed=11 + i, channels=3) for i in range(2)])
px = torch.from_numpy(frames).to("cuda:0").repeat((B + 1) // 2, 1, 1, 1)[:B].contiguous()
ex = oda.SurfExtractor(1920, 1080, 3, batch=B)
ex.set_timing(True)
acc = []
for i in range(10):
    ex.enqueue_batch(px.data_ptr(), B)
    ex.sync()
    acc.append(ex.stage_times())
acc = np.median(np.array(acc[2:]), axis=0)
print(" ".join("%s=%.3f" % (k, v) for k, v in zip(ex.STAGES, acc)), "total=%.3f ms" % acc.sum())
