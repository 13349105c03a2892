#!/usr/bin/env python3
"""Times the batched matcher on 16 pairs of n x n SIFT-like rows (device-side counts, capacity 27216)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import object_detection_amd as oda
from object_detection_amd import _lib
from synth import sift_like_descriptors
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3221
cap, B = int(os.environ.get('CAP', '27216')), int(os.environ.get('PAIRS', '16'))
DEVC = os.environ.get('DEVC', '1') == '1'
dev = torch.device("cuda", 0)
des = torch.zeros((B + 1, cap, 128), dtype=torch.float32, device=dev)
base = torch.from_numpy(sift_like_descriptors(n, 3)).to(dev)
for b in range(B + 1):
    des[b, :n] = torch.roll(base, b, 0)
cnt = torch.full((B + 1,), n, dtype=torch.int32, device=dev)
idx = torch.zeros((B, cap, 2), dtype=torch.int32, device=dev)
dist = torch.zeros((B, cap, 2), dtype=torch.float32, device=dev)
m = oda.Matcher(cap, cap, device=0, max_pairs=B)
pairs = [dict(q=des[b + 1].data_ptr(), db=des[b].data_ptr(), idx2=idx[b].data_ptr(), dist2=dist[b].data_ptr()) for b in range(B)]
if DEVC:
    for b, p in enumerate(pairs):
        p['nq'] = cnt[b + 1:].data_ptr(); p['nd'] = cnt[b:].data_ptr()
NQ = cap if DEVC else n
def step():
    m.top2_batch_device(pairs, NQ, NQ, int(os.environ.get('DIM','127')), 128, 0)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("cap=%d devc=%s n=%d split=%s  %.1f us/batch  %.1f TFLOP/s" % (cap, DEVC, n, os.environ.get("MIF_MATCH_SPLIT", "auto"), dt * 1e6, B * 2.0 * 127 * n * n / dt / 1e12))
