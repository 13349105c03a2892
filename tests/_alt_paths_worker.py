"""Runs one batched extraction and one batched match and stores the results in an .npz file. Started by
test_alt_paths_gpu.py once per setting of the library's diagnostic switches (they are read once per
process, so each setting needs a process of its own)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import object_detection_amd as oda  # noqa: E402
from synth import blob_image, sift_like_descriptors  # noqa: E402


def main(out):
    dev = torch.device("cuda:0")
    res = {}
    for tag, channels in (("gray", 1), ("bgr", 3)):
        frames = [blob_image(330, 215, seed=3 + i, channels=channels, shift=(2 * i, i)) for i in range(3)]
        px = torch.from_numpy(np.stack(frames)).to(dev)
        ex = oda.SiftExtractor(330, 215, channels, batch=3)
        ex.enqueue(px.data_ptr(), 3)
        for f in range(3):
            rc, feat = ex.copy_results(f)
            res["%s_rc%d" % (tag, f)] = np.int32(rc)
            res["%s_kp%d" % (tag, f)] = feat.keypoints
            res["%s_ori%d" % (tag, f)] = feat.orientations
            res["%s_des%d" % (tag, f)] = feat.descriptors
    # a BGR batch whose width is a multiple of four (the level-0 blur then fetches four pixels per load)
    frames = [blob_image(332, 216, seed=13 + i, channels=3, shift=(2 * i, i)) for i in range(3)]
    px = torch.from_numpy(np.stack(frames)).to(dev)
    ex = oda.SiftExtractor(332, 216, 3, batch=3)
    ex.enqueue(px.data_ptr(), 3)
    for f in range(3):
        rc, feat = ex.copy_results(f)
        res["bgr4_rc%d" % f] = np.int32(rc)
        res["bgr4_kp%d" % f] = feat.keypoints
        res["bgr4_des%d" % f] = feat.descriptors
    # SURF: points and the first and last response layer of a small BGR frame
    sx = oda.SurfExtractor(332, 216, 3)
    pts, raw = sx.run(frames[0])
    res["surf_points"] = pts.view(np.uint8)
    res["surf_raw"] = np.int32(raw)
    res["surf_integral"] = sx.integral()
    for layer in (0, 3, 7):
        r, l = sx.layer(layer)
        res["surf_resp%d" % layer], res["surf_lap%d" % layer] = r, l
    # a batch large enough for the screening passes (16 pairs of 1500 x 1400 rows)
    P, cap = 16, 1500
    Q = torch.zeros((P, cap, 128), dtype=torch.float32, device=dev)
    D = torch.zeros((P, cap, 128), dtype=torch.float32, device=dev)
    for p in range(P):
        q = sift_like_descriptors(cap, seed=900 + p)
        d = sift_like_descriptors(1400, seed=950 + p)
        d[7] = q[11]
        Q[p] = torch.from_numpy(q).to(dev)
        D[p, :1400] = torch.from_numpy(d).to(dev)
    idx = torch.zeros((P, cap, 2), dtype=torch.int32, device=dev)
    dist = torch.zeros((P, cap, 2), dtype=torch.float32, device=dev)
    m = oda.Matcher(cap, cap, max_pairs=P)
    pairs = [dict(q=Q[p].data_ptr(), db=D[p].data_ptr(), idx2=idx[p].data_ptr(), dist2=dist[p].data_ptr())
             for p in range(P)]
    m.top2_batch_device(pairs, cap, 1400, 128, 128, 0)
    torch.cuda.synchronize()
    res["match_idx"] = idx.cpu().numpy()
    res["match_dist"] = dist.cpu().numpy()
    np.savez(out, **res)


if __name__ == "__main__":
    main(sys.argv[1])
