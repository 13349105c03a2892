#!/usr/bin/env python3
"""Descriptor matches per second of the brute-force top-2 matcher on one GPU.

Queries and database are seeded SIFT-like rows already resident in HBM; a step
resolves every query row against the whole database (squared-L2 GEMM on the
matrix cores + running top-2 + exact distances of the winners). Prints one JSON
line with matches/s and the matrix-core utilisation (2 * dim * nq * nd flop).

python tools/bench_match.py --nq 65536 --nd 262144 --steps 3
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X dense fp32 matrix peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=65536)
    ap.add_argument("--nd", type=int, default=262144)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()
    import numpy as np
    import torch
    import object_detection_amd as oda
    from synth import sift_like_descriptors
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(7)
    base = sift_like_descriptors(min(args.nd, 65536), seed=7, dim=args.dim)
    reps = (args.nd + len(base) - 1) // len(base)
    db = np.concatenate([np.roll(base, r, axis=1) for r in range(reps)])[:args.nd]
    perm = rng.permutation(args.nd)[:args.nq] if args.nq <= args.nd else rng.integers(0, args.nd, args.nq)
    q = np.clip(db[perm] + rng.normal(0, 8, (args.nq, args.dim)).astype(np.float32), 0, 255).astype(np.float32)
    tq = torch.from_numpy(np.ascontiguousarray(q)).to(dev)
    td = torch.from_numpy(np.ascontiguousarray(db)).to(dev)
    m = oda.Matcher(args.nq, args.nd, device=0)
    idx = torch.zeros((args.nq, 2), dtype=torch.int32, device=dev)
    dist = torch.zeros((args.nq, 2), dtype=torch.float32, device=dev)

    def step():
        m.top2_device(tq.data_ptr(), args.nq, td.data_ptr(), args.nd, args.dim, args.dim, idx.data_ptr(),
                      dist.data_ptr())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    hit = float((idx[:, 0].cpu().numpy() == perm).mean()) if args.nq <= args.nd else None
    flops = 2.0 * args.dim * args.nq * args.nd
    print(json.dumps({"metric": "descriptor matches/sec", "value": args.nq / dt, "unit": "matches/s",
                      "nq": args.nq, "nd": args.nd, "dim": args.dim, "ms_per_step": dt * 1e3,
                      "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": flops / dt / 1e12 / MFMA_F32_PEAK_TFLOPS},
                      "true_neighbour_recovered": hit}))


if __name__ == "__main__":
    main()
