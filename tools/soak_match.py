#!/usr/bin/env python3
"""Randomised parity soak of the matcher: the bf16-screened search against the plain fp32 search (same
handle, screening switched off) on random problem sizes, pair counts, dimensions and row contents (SIFT-like
rows, many copies of a few rows, near-copies of queries, scaled values); indices and distances must be
bit-identical. The library's switches are read once per process: run it once per setting, e.g. with
MIF_MATCH_SCREEN_TURNS=1 MIF_MATCH_SAMPLE=3. Not part of the test suite.
usage: python tools/soak_match.py --seconds 120 [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import object_detection_amd as oda  # noqa: E402
from synth import sift_like_descriptors  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(args.seed)
    pool = sift_like_descriptors(40000, seed=args.seed)
    t0, cases, bad, queries = time.time(), 0, 0, 0
    while time.time() - t0 < args.seconds:
        shape = rng.choice(["batch", "single", "wide"])
        if shape == "batch":
            P, nq, nd = int(rng.randint(2, 40)), int(rng.randint(200, 3000)), int(rng.randint(130, 3000))
        elif shape == "single":
            P, nq, nd = 1, int(rng.randint(500, 20000)), int(rng.randint(2000, 40000))
        else:
            P, nq, nd = 1, int(rng.randint(16384, 40000)), int(rng.randint(20000, 40000))
        dim = int(rng.choice([128, 128, 127, 64]))
        style = rng.choice(["plain", "copies", "near", "scaled"])
        Q = torch.zeros((P, nq, 128), dtype=torch.float32, device=dev)
        D = torch.zeros((P, nd, 128), dtype=torch.float32, device=dev)
        for p in range(P):
            q = pool[rng.randint(0, len(pool), nq)].copy()
            if style == "copies":
                d = np.repeat(pool[rng.randint(0, len(pool), 5)], (nd + 4) // 5, axis=0)[:nd][rng.permutation(nd)]
            else:
                d = pool[rng.randint(0, len(pool), nd)].copy()
            if style == "near":
                k = min(nq, nd, 300)
                d[:k] = q[:k]
                d[np.arange(k), rng.randint(0, 128, k)] += 1.0
            if style == "scaled":
                s = np.float32(rng.choice([1e-3, 1e3]))
                q, d = q * s, d * s
            Q[p] = torch.from_numpy(np.ascontiguousarray(q)).to(dev)
            D[p] = torch.from_numpy(np.ascontiguousarray(d)).to(dev)
        if dim < 128:
            Q[:, :, dim:] = 0
            D[:, :, dim:] = 0
        m = oda.Matcher(nq, nd, max_pairs=P)
        res = []
        for screening in (True, False):
            m.set_screening(screening)
            idx = torch.full((P, nq, 2), -9, dtype=torch.int32, device=dev)
            dist = torch.zeros((P, nq, 2), dtype=torch.float32, device=dev)
            pairs = [dict(q=Q[p].data_ptr(), db=D[p].data_ptr(), idx2=idx[p].data_ptr(), dist2=dist[p].data_ptr())
                     for p in range(P)]
            m.top2_batch_device(pairs, nq, nd, dim, 128, 0)
            torch.cuda.synchronize()
            res.append((idx.cpu().numpy(), dist.cpu().numpy(), m.last_screen_passes()))
        ok = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))
        if not ok:
            bad += 1
            print("MISMATCH: pairs %d nq %d nd %d dim %d %s passes %.3f" % (P, nq, nd, dim, style, res[0][2]), flush=True)
        cases += 1
        queries += P * nq
        if cases % 20 == 0:
            print("%d cases, %d queries, %d bad, %.0f s (last: %s %s passes %.3f)"
                  % (cases, queries, bad, time.time() - t0, shape, style, res[0][2]), flush=True)
    print("done: %d cases, %d queries, %d bad" % (cases, queries, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
