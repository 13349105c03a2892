"""Time of the integral-image stage of a SURF batch (HIP events at the stage boundary), several repeats.
usage: python3 tools/time_integral.py [batch] [width height]"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import object_detection_amd as oda

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
rng = np.random.default_rng(3)
px = torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)).to("cuda:0")
ex = oda.SurfExtractor(W, H, 3, batch=B)
ex.set_timing(True)
ts = []
for i in range(12):
    ex.enqueue_batch(px.data_ptr(), B)
    ex.sync()
    ts.append(ex.stage_times()[0])
ts = sorted(ts[2:])
print("integral stage ms: min %.4f median %.4f  (%d frames %dx%d, %.2f TB/s on 7 B/px at the median)"
      % (ts[0], ts[len(ts) // 2], B, W, H, B * W * H * 7 / ts[len(ts) // 2] * 1e-9))
