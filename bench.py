#!/usr/bin/env python3
"""1080p SIFT extract + match throughput on MI355X (the metric of BASELINE.json).

A step = one batch of `--batch` synthetic 1920x1080 BGR frames, already resident
in HBM, through the whole hot path: gray conversion, 3-octave scale space + DoG,
detection, orientation, 128-d descriptors, then brute-force matching of every
frame against the previous frame (frame 0 of a batch against the last frame of
the previous batch) with the reference's SIFT good-match rule. Frames are
independent, so ranks process different frames with no data-path collective
(weak scaling); value = all ranks' frames / max-over-ranks time.

python bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
# Algorithmic bytes of the blur launches per pixel and octave: horizontal blur 6 x (r4 + w4), vertical
# blur 6 x (r4 + w4) and the base image (gray conversion or decimation, r4 + w4), which the level-0 launch
# forms while it stages its window = 104 B, spread over the six blur_level launches of an octave. The DoG
# images (another 40 B in the reference's dataflow) are not part of these launches: they are never stored,
# the detector subtracts neighbouring levels as it reads them.
ALG_BYTES_PER_PX_OCTAVE = 104.0


def min_bytes_per_frame(px, channels):
    """What the fused blur_level launches have to move at least: r4 + w4 per level, except that level 0
    reads the frame's bytes (octave 0) or every other pixel of every other row of the previous octave's
    level 3 (octaves 1, 2: 8 bytes fetched per pixel, memory moves whole sectors)."""
    return 44.0 * sum(px) + channels * px[0] + 8.0 * sum(px[1:])



def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512,
                    help="frames per step; two handles of 0.147 GB per 1080p frame stay resident (151 GB at 512)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--channels", type=int, default=3, choices=[1, 3])
    ap.add_argument("--no-match", action="store_true",
                    help="extraction only (the 4K batch configuration: --width 3840 --height 2160 --channels 1)")
    ap.add_argument("--host-frames", action="store_true",
                    help="frames start in pinned host memory: every step uploads its batch over PCIe on a copy "
                         "stream (overlapping the previous batch) and downloads the match summaries; the "
                         "PCIe-inclusive rate, never the headline value")
    ap.add_argument("--native-pipeline", action="store_true",
                    help="drive the same schedule through the library's own pipeline object (mif_video_*) "
                         "instead of the calls made from this script; no per-stage times in that mode")
    ap.add_argument("--serial", action="store_true",
                    help="run the matcher of a batch after its feature phase instead of beside the next one's")
    return ap.parse_args()


def _frame(job):
    from synth import blob_image
    width, height, seed, t, channels = job
    return blob_image(width, height, seed=seed, channels=channels, shift=(2 * t, t), wrap=True)


def make_frames(width, height, batch, rank, channels=3):
    """Frame t = the seeded blob field translated by (2t, t), wrapping around the frame so that the
    blob count (and with it the keypoint count, about 3000) is the same for every t. Rendered by a few worker processes
    (numpy only; called before anything touches the GPU)."""
    jobs = [(width, height, 1234 + 100 * rank, t, channels) for t in range(batch)]
    workers = min(8, batch)
    if workers > 1:
        import multiprocessing as mp
        # close + join, not the context manager: its exit terminates the workers with SIGTERM,
        # which a profiler's signal handler inherited by the workers may turn into a hang
        pool = mp.get_context("fork").Pool(workers)
        try:
            frames = pool.map(_frame, jobs)
        finally:
            pool.close()
            pool.join()
    else:
        frames = [_frame(j) for j in jobs]
    return np.stack(frames)


class Pipeline:
    """Two batched SIFT handles used alternately (own stream each), one matcher on a stream of its own.

    A step extracts batch k and matches batch k - 1 (every frame against the previous frame; frame 0
    against the last frame of batch k - 2, whose descriptors were copied aside when that batch
    finished). Order on the GPU:

        pyramid(k)  ->  features(k)  ||  matcher(k - 1)   ->  pyramid(k + 1)  -> ...

    The scale-space phase (bandwidth kernels) runs alone; the feature phase (vector and latency
    bound) shares the GPU with the matcher (matrix cores). With serial=True the matcher of batch k
    runs after features(k) instead and nothing overlaps."""

    def __init__(self, oda, torch, dev_index, width, height, batch, serial=False, channels=3, match=True):
        from object_detection_amd import _lib
        self._lib, self.lib = _lib, oda.load_library()
        self.serial = serial
        self.oda, self.torch, self.B, self.dev_index = oda, torch, batch, dev_index
        self.dev = torch.device("cuda", dev_index)
        self.match = match
        self.h = [oda.SiftExtractor(width, height, channels, device=dev_index, batch=batch) for _ in range(2)]
        self.cap = self.h[0].capacity
        self.m = oda.Matcher(self.cap, self.cap, device=dev_index, max_pairs=batch)
        self.mstream = torch.cuda.Stream(device=self.dev)
        B, cap = batch, self.cap
        # results of the matcher per handle parity
        self.idx = [torch.zeros((B, cap, 2), dtype=torch.int32, device=self.dev) for _ in range(2)]
        self.dist = [torch.zeros((B, cap, 2), dtype=torch.float32, device=self.dev) for _ in range(2)]
        self.good = [torch.zeros((B, cap), dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self.summ = [torch.zeros((B, 2), dtype=torch.int32, device=self.dev) for _ in range(2)]
        # descriptors and count of the last frame of a finished batch (three in rotation: one is
        # being written while the matcher may still read the one from two batches ago)
        self.snap_des = [torch.zeros((cap, 128), dtype=torch.float32, device=self.dev) for _ in range(3)]
        self.snap_cnt = [torch.zeros((1,), dtype=torch.int32, device=self.dev) for _ in range(3)]
        self.res = [h.device_results() for h in self.h]
        self._tables = {}
        self.step_no = 0
        self.pending = False

    def _pairs(self, k):
        """Pair table of batch k: queries in handle k % 2, database of frame 0 in snapshot (k - 1) % 3."""
        key = (k % 2, (k - 1) % 3)
        if key not in self._tables:
            B, cap = self.B, self.cap
            _, _, des, cnt = self.res[k % 2]
            out = []
            for b in range(B):
                if b == 0:
                    db, nd = self.snap_des[(k - 1) % 3].data_ptr(), self.snap_cnt[(k - 1) % 3].data_ptr()
                else:
                    db, nd = des + (b - 1) * cap * 512, cnt + (b - 1) * 32 + 12
                out.append(dict(q=des + b * cap * 512, db=db, nq=cnt + b * 32 + 12, nd=nd,
                                idx2=self.idx[k % 2][b].data_ptr(), dist2=self.dist[k % 2][b].data_ptr(),
                                good=self.good[k % 2][b].data_ptr(), summary=self.summ[k % 2][b].data_ptr()))
            self._tables[key] = out
        return self._tables[key]

    def _wait(self, waiter, signaller):
        self._lib.check(self.lib.mif_stream_wait_stream(self.dev_index, ctypes.c_void_p(waiter),
                                                        ctypes.c_void_p(signaller)))

    def _match(self, k):
        if not self.match:
            return
        self.m.top2_batch_device(self._pairs(k), self.cap, self.cap, 127, 128, self._lib.FILTER_SIFT,
                                 stream=self.mstream.cuda_stream)

    def step(self, frames_ptr):
        k = self.step_no
        h, other, ms = self.h[k % 2], self.h[(k - 1) % 2], self.mstream.cuda_stream
        self._wait(h.stream(), other.stream())     # extraction runs batch after batch
        self._wait(h.stream(), ms)                 # the matcher two batches back read this handle's results
        h.enqueue_pyramid(frames_ptr, self.B)
        if self.pending and not self.serial:
            self._wait(ms, h.stream())             # beside the feature phase, not the scale spaces
            self._match(k - 1)
        h.enqueue_features(True)
        h.snapshot_frame(self.B - 1, self.snap_des[k % 3].data_ptr(), self.snap_cnt[k % 3].data_ptr())
        self.pending = True
        self.step_no += 1
        if self.serial:
            self.flush()

    def flush(self):
        """Matches the batch extracted last, if that has not happened yet."""
        if self.pending:
            k = self.step_no - 1
            self._wait(self.mstream.cuda_stream, self.h[k % 2].stream())
            self._match(k)
            self.pending = False

    def last(self):
        return (self.step_no - 1) % 2

    def sync(self):
        self.flush()
        for h in self.h:
            h.sync()
        self.mstream.synchronize()


def cpu_baseline(frames, budget_s, match=True):
    """CPU restatement of the reference's -f 0 path (oracle/, OpenMP) on a bounded sample of the same
    frames: extraction of every sampled frame plus matching against the previous sampled frame."""
    import oracle_lib as orc
    import object_detection_amd as oda
    n, h, w = frames.shape[:3]
    S = orc.SiftOracle(w, h, 1 if frames.ndim == 3 else frames.shape[3])
    prev = None
    done, t0 = 0, time.perf_counter()
    while done < max(2, min(n, 64)):
        S.run(frames[done % n])
        des = S.all_features()[3]
        if match and prev is not None and len(des) and len(prev):
            idx, dist = orc.match_top2(oda.pack_sift_rows(des), oda.pack_sift_rows(prev))
            mn = min(100.0, float(dist[:, 0].min()))
            _ = dist[:, 0] <= max(2.5 * mn, 0.35)
        prev = des
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": orc.lib().orc_get_threads(), "kind": "port",
            "sample": "%d frames %dx%d %s, %s, OpenMP restatement of the reference -f 0 path (oracle/)"
                      % (done, w, h, "gray" if frames.ndim == 3 else "BGR",
                         "extract + match vs previous frame" if match else "extraction only")}


def run_native(args, oda, torch, frames, local_rank, world, rank, dist):
    """The timed loop over mif_video_submit_device / mif_video_collect (same schedule, made inside the
    library); prints a reduced line (no roofline: the stage events belong to the handles inside)."""
    vp = oda.VideoPipeline(args.width, args.height, args.channels, batch=args.batch, device=local_rank)
    for _ in range(args.warmup):
        vp.submit_device(frames.data_ptr(), args.batch)
    vp.collect()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vp.submit_device(frames.data_ptr(), args.batch)
    res = vp.collect()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=frames.device if dist is not None and dist.get_backend() == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "1080p SIFT extract+match frames/sec", "value": world * args.steps * args.batch / dt,
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "native pipeline object (mif_video_*), same schedule",
                                     "frames_per_step": args.batch,
                                     "keypoints_last_frame": list(res[-1].keypoints),
                                     "good_matches_last_frame": res[-1].good_matches}}))


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    frames_np = make_frames(args.width, args.height, args.batch, rank, args.channels)      # before the GPU is initialised
    import torch
    import object_detection_amd as oda
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # one rank per GPU; BENCH_BACKEND=gloo lets several ranks share a GPU for a rehearsal of this path
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    frames = torch.from_numpy(frames_np).to(torch.device("cuda", local_rank))
    if args.native_pipeline:
        return run_native(args, oda, torch, frames, local_rank, world, rank, dist)
    pipe = Pipeline(oda, torch, local_rank, args.width, args.height, args.batch, serial=args.serial,
                    channels=args.channels, match=not args.no_match)

    if args.host_frames:
        host = torch.from_numpy(frames_np).pin_memory()
        dev_buf = [torch.empty_like(frames), frames]
        ncopy = max(1, min(int(os.environ.get("BENCH_COPY_STREAMS", "4")), args.batch))
        cstreams = [torch.cuda.Stream(device=frames.device) for _ in range(ncopy)]
        dstream = torch.cuda.Stream(device=frames.device)
        cuts = [args.batch * i // ncopy for i in range(ncopy + 1)]
        host_summ = [torch.empty((args.batch, 2), dtype=torch.int32).pin_memory() for _ in range(2)]

        def step():
            k = pipe.step_no
            hk = pipe.h[k % 2]
            # buffer k % 2 was last read by the scale spaces of batch k - 2, enqueued on this handle's stream
            # (several copy streams: one DMA engine alone does not fill the link)
            for ci, cs in enumerate(cstreams):
                pipe._wait(cs.cuda_stream, hk.stream())
                with torch.cuda.stream(cs):
                    dev_buf[k % 2][cuts[ci]:cuts[ci + 1]].copy_(host[cuts[ci]:cuts[ci + 1]], non_blocking=True)
            for cs in cstreams:
                pipe._wait(hk.stream(), cs.cuda_stream)
            pipe.step(dev_buf[k % 2].data_ptr())
            if k >= 1 and pipe.match and not pipe.serial:
                # the matcher of batch k - 1 has just been enqueued: its summaries go back to the host
                pipe._wait(dstream.cuda_stream, pipe.mstream.cuda_stream)
                with torch.cuda.stream(dstream):
                    host_summ[(k - 1) % 2].copy_(pipe.summ[(k - 1) % 2], non_blocking=True)
    else:
        def step():
            pipe.step(frames.data_ptr())

    for _ in range(args.warmup):
        step()
    pipe.sync()
    for h in pipe.h:
        h.set_timing(True)

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    pipe.sync()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0

    t = torch.tensor([dt], dtype=torch.float64, device=frames.device if backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    # stage times of the last two batches of the timed region (one per handle), measured with HIP events on
    # the handles' own streams; blur_level is the dominant bandwidth kernel (6 launches per octave)
    timed = [pipe.h[(pipe.last() - i) % len(pipe.h)] for i in range(min(len(pipe.h), args.steps))]
    stage = np.mean([h.stage_times() for h in timed], axis=0)             # ms [octave][stage]
    px = [h_ * w_ for (w_, h_, _) in (pipe.h[0].octave_info(o) for o in range(3))]
    blur_ms = float(stage[:, 1].sum())
    alg_bytes = ALG_BYTES_PER_PX_OCTAVE * sum(px) * args.batch
    min_bytes = min_bytes_per_frame(px, args.channels) * args.batch
    achieved = alg_bytes / (blur_ms * 1e-3) / 1e9
    launches = 18
    # HBM-side bytes of the same launches from the PMC passes (FETCH_SIZE and WRITE_SIZE, separate
    # rocprofv3 runs of this script, summarised by tools/pmc_traffic.py); they scale with the frame count
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_blur_traffic.json")) as fh:
            pmc = json.load(fh)
        if args.width == 1920 and args.height == 1080:
            traffic = pmc["blur_bytes_per_frame"] * args.batch / launches
    except (OSError, KeyError, ValueError):
        pass
    summ = pipe.summ[pipe.last()].cpu().numpy()
    counts = pipe.h[pipe.last()].read_counts(args.batch - 1)

    # yardstick measured on this GPU: a 1 GiB device-to-device copy (bytes read + bytes written)
    src_t = torch.empty(1 << 28, dtype=torch.float32, device=frames.device)
    dst_t = torch.empty_like(src_t)
    dst_t.copy_(src_t)
    torch.cuda.synchronize()
    tc = time.perf_counter()
    for _ in range(5):
        dst_t.copy_(src_t)
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * src_t.numel() * 4 / (time.perf_counter() - tc) / 1e9
    del src_t, dst_t

    if rank == 0:
        frames_total = args.batch * args.steps * world
        out = {
            "metric": "1080p SIFT extract+match frames/sec"
                      if (args.width, args.height, args.no_match) == (1920, 1080, False)
                      else "%dx%d SIFT %s frames/sec" % (args.width, args.height,
                                                          "extraction" if args.no_match else "extract+match"),
            "value": frames_total / dt_max,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" + (", uploaded from pinned host memory every step" if args.host_frames else ""),
            "config": {"workload": ("SIFT extract + match-vs-previous-frame on a synthetic %dx%d BGR sequence "
                                    "(stand-in for basketball.mp4, BASELINE.json configs[2])" % (args.width, args.height))
                       if (not args.no_match and args.channels == 3) else
                       ("SIFT %s on a synthetic %dx%dx%d sequence" % ("extraction" if args.no_match else
                                                                       "extract + match", args.width, args.height,
                                                                       args.channels)),
                       "frames_per_step": args.batch, "octaves": 3, "levels": 6,
                       "pipeline": "matcher after each batch" if args.serial else
                                   "matcher of batch k-1 beside the feature phase of batch k",
                       "keypoints_last_frame": counts[1], "good_matches_last_frame": int(summ[-1, 0]),
                       "parallelism": "frames sharded over %d GPU(s), no collective" % world},
            "roofline": {
                "bound": "hbm", "kernel": "k_blur_level (base image + H + V fused, 18 launches per batch)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "avg_launch_us": blur_ms * 1e3 / launches,
                "alg_bytes_per_launch": alg_bytes / launches,
                "min_bytes_per_launch": min_bytes / launches,
                "achieved_min_traffic": achieved * min_bytes / alg_bytes,
                "achieved_traffic": (traffic / (blur_ms * 1e-3 / launches) / 1e9) if traffic else None,
                "measured_copy_GBps": copy_gbs,
            },
            "stage_ms_per_batch": {"base": float(stage[:, 0].sum()), "scale_space": blur_ms,
                                   "detector": float(stage[:, 2].sum()), "orient_desc": float(stage[:, 3].sum())},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames_np, args.cpu_seconds, match=not args.no_match)
        print(json.dumps(out), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
