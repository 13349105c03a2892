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

With --gpus N > 1 and no WORLD_SIZE in the environment this process only starts N ranks of itself
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one per GPU), forwards rank 0's JSON line and exits
non-zero if any rank does; under `python -m torch.distributed.run` the ranks exist already and every
process is a rank.

The one JSON line carries, beside the frames/s headline and its roofline block:
  matcher             descriptor matches/s of the 1 M x 1 M x 128 search (BASELINE.json configs[4]), exact
                      fp32 and bf16-screened, against the matrix-core peaks; with N ranks the database rows
                      are sharded and the top-2 candidates gathered over RCCL by the library itself
  value_host_to_host  the same step with the frames starting in pinned host memory and the match
                      summaries ending there (PCIe included; never the headline value)
  value_host_to_host_features
                      the same, and every batch's keypoints, orientations and descriptors also land in pinned
                      host memory (what SIFT::extract_features hands back, SIFT/sift.cpp:1889-1911), on a
                      download stream beside the next batch's work
  configs             the other configurations of BASELINE.json, each bounded to a few seconds and compared
                      with the CPU checker: C1 SIFT box.pgm / scene.pgm (extract + match + homography, host
                      to host, through the C++ demo), C2 the same pair through SURF plus a 1080p batch,
                      C4 a batch of 4K single-channel frames; each with the CPU restatement timed beside it
  verified            frame 1 of the last timed batch compared with the CPU checker (oracle/): keypoint
                      counts per octave, good-match count and the matched indices; exit code 1 on mismatch
  per_rank_ms_per_step  every rank's own time per step (the headline uses the slowest)
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


_T0 = time.perf_counter()


def progress(msg):
    """One line on stderr per finished part of the run (rank 0): stdout carries the JSON line only."""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %6.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak
MFMA_F32_PEAK_TFLOPS = 157.3     # dense fp32 matrix peak
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 matrix peak
# What the reference's dataflow moves for the same launches, per pixel and octave: horizontal blur
# 6 x (r4 + w4), vertical blur 6 x (r4 + w4) and the base image (gray conversion or decimation, r4 + w4)
# = 104 B. Reported as `alg_unfused`; the roofline fraction itself uses what the fused launches have to
# move (min_bytes_per_frame), so that it is a utilisation and stays below 1.
ALG_UNFUSED_BYTES_PER_PX_OCTAVE = 104.0
BLUR_TRAFFIC_FILE = os.path.join("profiles", "r03_blur_traffic.json")
SURF_TRAFFIC_FILE = os.path.join("profiles", "r03_surf_traffic.json")
BLUR_4K_TRAFFIC_FILE = os.path.join("profiles", "r03_4k_blur_traffic.json")
# integral image of a BGR frame: 3 bytes read, 4 written (both passes in one launch). The unit of a scan per
# pass, r4 + w4 twice = 16 B/px, is kept beside it (a launch per pass moves 15, the reference with its
# transposes about 32).
SURF_INTEGRAL_BYTES_PER_PX = 7.0
SURF_INTEGRAL_UNFUSED_BYTES_PER_PX = 16.0
GOLDEN = os.path.join(ROOT, "tests", "golden")
DEMO_DIR = os.path.join(ROOT, "object_detection_amd", "cpp", "build")


def min_bytes_per_frame(px, channels):
    """Algorithmic bytes of the scale-space launches of one frame: every level is read once and written
    once (r4 + w4 per pixel), except that level 0 reads the frame's bytes (octave 0) or every other pixel
    of every other row of the previous octave's level 3 (octaves 1, 2: 8 bytes fetched per pixel, memory
    moves whole sectors). Launches that build two levels at once move less than this (the first level of
    the pair is not read back): the counters' `traffic` shows it, the algorithmic figure stays."""
    return 44.0 * sum(px) + channels * px[0] + 8.0 * sum(px[1:])


def parse_args(argv=None):
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
                    help="make the PCIe-inclusive schedule the timed region of the headline fields too (frames "
                         "start in pinned host memory, summaries end there); for profiling that schedule")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the value_host_to_host measurement")
    ap.add_argument("--no-two-streams-leg", action="store_true", help="skip the value_two_streams measurement")
    ap.add_argument("--host-steps", type=int, default=8, help="timed steps of the host-to-host legs")
    ap.add_argument("--no-feature-leg", action="store_true", help="skip the value_host_to_host_features measurement")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` block (SIFT / SURF box-scene pair, SURF 1080p batch, 4K batch)")
    ap.add_argument("--config-batch", type=int, default=64, help="frames per step of the batched `configs` legs")
    ap.add_argument("--config-steps", type=int, default=5, help="timed steps of the batched `configs` legs")
    ap.add_argument("--pair-repeats", type=int, default=30, help="timed repeats of the box / scene legs")
    ap.add_argument("--match-rows", type=int, default=1000000,
                    help="query rows = database rows of the matcher block (0 = skip it)")
    ap.add_argument("--match-steps", type=int, default=2, help="timed passes per form of the matcher block")
    ap.add_argument("--match-timeout", type=int, default=300,
                    help="seconds after which a multi-rank matcher block is given up (the line is printed without it)")
    ap.add_argument("--deadline", type=int, default=540,
                    help="seconds after the start at which rank 0 prints the line as far as it has got (the frame "
                         "pipeline's figures are complete by then, later blocks may be missing; exit code 4) instead "
                         "of running on; 0 = no deadline")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparison with the CPU checker")
    ap.add_argument("--native-pipeline", action="store_true",
                    help="drive the same schedule through the library's own pipeline object (mif_video_*) "
                         "instead of the calls made from this script; no per-stage times in that mode")
    ap.add_argument("--serial", action="store_true",
                    help="run the matcher of a batch after its feature phase instead of beside the next one's")
    ap.add_argument("--launch-only", action="store_true",
                    help="start the ranks, form the process group (gloo), report who is there and stop: "
                         "checks the launcher on a machine without GPUs")
    return ap.parse_args(argv)


# ------------------------------------------------------------------ launcher

def launch_ranks(args, argv):
    """Parent of an N-rank run: never touches the GPU. Starts N copies of this script, one per GPU,
    with the environment `torch.distributed.run` would give them; rank 0 prints the JSON line on the
    stdout they all inherit. Returns the exit code (the first non-zero one, after stopping the rest)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n = args.gpus
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    live = set(range(n))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print("bench.py: rank %d exited with %d; stopping the other ranks" % (r, code), file=sys.stderr)
                for o in live:
                    procs[o].terminate()            # exactly the processes started above
        if live:
            time.sleep(0.05)
    return rc


def gather_per_rank(dist, value):
    """[value of rank 0, value of rank 1, ...] on every rank (a list of one without a process group)."""
    if dist is None:
        return [value]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, value)
    return out


def run_launch_only(rank, world, local_rank):
    """The process group without the GPU work: every rank reports its device index and a time of its own,
    rank 0 prints them in the fields the full run uses (config.rank_devices, per_rank_ms_per_step,
    matcher.sharded), so that the launcher and the line's multi-rank fields can be checked on a machine
    without GPUs."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend="gloo")
    group = dist if world > 1 else None
    devs = gather_per_rank(group, local_rank)
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
    per_rank = gather_per_rank(group, (time.perf_counter() - t0) * 1e3)
    if rank == 0:
        print(json.dumps({"launch_only": True, "n_gpus": world,
                          "frame_workers": frame_workers(512, world),
                          "per_rank_ms_per_step": per_rank,
                          "config": {"world_size": dist.get_world_size() if world > 1 else 1,
                                     "rank_devices": devs},
                          # the full run fills this from the library's handle (mif_match_sharded_ranks)
                          "matcher": {"sharded": {"nranks": len(per_rank), "nlocal": 1, "uses_rccl": None,
                                                  "source": "launch-only: no GPU, no library handle"}}}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# -------------------------------------------------------------------- frames

def _frame(job):
    from synth import blob_image
    width, height, seed, t, channels = job
    return blob_image(width, height, seed=seed, channels=channels, shift=(2 * t, t), wrap=True)


def frame_workers(batch, world):
    """Worker processes one rank renders its frames with: at most 8, and all ranks of a node together no
    more than the machine has CPUs (every rank renders at the same time)."""
    return max(1, min(8, batch, (os.cpu_count() or 1) // max(world, 1)))


def make_frames(width, height, batch, rank, channels=3, world=1, seed=1234):
    """Frame t = the seeded blob field translated by (2t, t), wrapping around the frame so that the
    blob count (and with it the keypoint count, about 3000) is the same for every t. Rendered by a few worker processes
    (numpy only; called before anything touches the GPU)."""
    jobs = [(width, height, seed + 100 * rank, t, channels) for t in range(batch)]
    workers = frame_workers(batch, world)
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
    """Two batched SIFT handles used alternately and one matcher, all on ONE stream.

    A step extracts batch k and matches batch k - 1 (every frame against the previous frame; frame 0
    against the last frame of batch k - 2, whose descriptors were copied aside when that batch
    finished). Order on the GPU:

        pyramid(k)  ->  matcher(k - 1)  ->  features(k)  ->  pyramid(k + 1)  -> ...

    Every kernel of the chain fills the GPU, so a matcher stream of its own (BENCH_MATCH_STREAM=1: the
    matcher beside the feature phase) gains nothing, and it costs uploads their overlap: a stream that
    waits for an event blocks the hardware queue it shares with other streams (a process has 4), and
    the copy stream's commands ended up behind the matcher's wait for pyramid(k). With serial=True the
    matcher of batch k runs right after features(k)."""

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
        self.split = int(os.environ.get("BENCH_CU_SPLIT", "256" if os.environ.get("BENCH_TWO_STREAMS") else "0"))
        self.masked = []
        if self.split:
            # experiment: the two handles on streams that own disjoint shares of the compute units
            # (the low `split` of 256 mask bits and the rest; 256 = BENCH_TWO_STREAMS: both streams own the
            # whole device and run their batches side by side), each with a matcher of its own
            for lo, hi in ((0, self.split), (self.split, 256)) if self.split < 256 else ((0, 256), (0, 256)):
                words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if lo <= 32 * wd + b < hi)
                                                for wd in range(8)])
                st = ctypes.c_void_p()
                self._lib.check(self.lib.mif_stream_create_masked(dev_index, words, 8, ctypes.byref(st)))
                self.masked.append(st.value)
            self.created_streams = list(self.masked)
            if os.environ.get("BENCH_CU_SAME"):        # both handles on the first share: what a share can do alone
                self.masked[1] = self.masked[0]
            self.h[0].set_stream(self.masked[0])
            self.h[1].set_stream(self.masked[1])
            self.m2 = [self.m, oda.Matcher(self.cap, self.cap, device=dev_index, max_pairs=batch)]
            self.serial = True
        else:
            self.h[1].set_stream(self.h[0].stream())
        if self.split:
            self.mstream = torch.cuda.ExternalStream(self.masked[0], device=self.dev)
        elif os.environ.get("BENCH_MATCH_STREAM"):
            self.mstream = torch.cuda.Stream(device=self.dev)
        else:
            self.mstream = torch.cuda.ExternalStream(self.h[0].stream(), device=self.dev)
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

    def enable_two_streams(self):
        """From now on the two handles run on a stream each (both own the whole device) with a matcher
        each, batch k's matcher waiting for batch k - 1's last frame: the kernels of two batches overlap."""
        self.sync()
        self.h[1].set_stream(None)
        self.masked = [self.h[0].stream(), self.h[1].stream()]
        self.m2 = [self.m, self.oda.Matcher(self.cap, self.cap, device=self.dev_index, max_pairs=self.B)]
        self.split, self.serial = 256, True
        self.mstream = self.torch.cuda.ExternalStream(self.masked[0], device=self.dev)

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
        if self.split:
            st, other = self.masked[k % 2], self.masked[(k - 1) % 2]
            self._wait(st, other)                  # the previous batch's last frame has been copied aside
            self.m2[k % 2].top2_batch_device(self._pairs(k), self.cap, self.cap, 127, 128, self._lib.FILTER_SIFT,
                                             stream=st)
            return
        self.m.top2_batch_device(self._pairs(k), self.cap, self.cap, 127, 128, self._lib.FILTER_SIFT,
                                 stream=self.mstream.cuda_stream)

    def step(self, frames_ptr):
        k = self.step_no
        h, other, ms = self.h[k % 2], self.h[(k - 1) % 2], self.mstream.cuda_stream
        # (waits between equal streams return at once: stream order already holds)
        if not self.split:
            self._wait(h.stream(), other.stream())     # extraction runs batch after batch
            self._wait(h.stream(), ms)                 # the matcher two batches back read this handle's results
        h.enqueue_pyramid(frames_ptr, self.B)
        if self.pending and not self.serial:
            self._wait(ms, h.stream())             # after the scale spaces
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
            if not self.split:
                self._wait(self.mstream.cuda_stream, self.h[k % 2].stream())
            self._match(k)
            self.pending = False

    def last(self):
        return (self.step_no - 1) % 2

    def close(self):
        """Detaches the handles from the streams this object created for them (mif_stream_create_masked)
        and destroys those streams; the handles go back to their own streams, as the stream's lifetime
        contract asks (a caller's stream must outlive its use by the handle or be detached first)."""
        self.sync()
        for h in self.h:                # (handle 1 runs on handle 0's stream in the default schedule)
            h.set_stream(None)
        for st in getattr(self, "created_streams", []):
            self._lib.check(self.lib.mif_stream_destroy(self.dev_index, ctypes.c_void_p(st)))
        self.created_streams = []

    def sync(self):
        self.flush()
        for h in self.h:
            h.sync()
        self.mstream.synchronize()


class FeatureDownload:
    """What SIFT::extract_features hands back (SIFT/sift.cpp:1889-1911: keypoints, orientations and
    descriptors in the caller's memory) for whole batches: after the feature phase of batch k the rows of all
    its frames are packed back to back on the device (mif_sift_pack_results) and their offsets downloaded
    on the compute stream (2 KB); once the host knows the row count it enqueues three copies of exactly
    that many rows into pinned host memory on a download stream, which run beside the next batches'
    kernels. The host asks for the row count of batch k - 2 when it has enqueued batch k, so it does not
    wait in practice and keeps enqueuing uploads and kernels a batch ahead of the GPU. Device staging and
    pinned buffers exist twice and alternate; the packing of batch k waits (on the device) for the
    download of batch k - 2. The download stream never waits for an event, so it cannot hold up a stream
    that shares its hardware queue."""

    def __init__(self, pipe, torch, device, rows_per_frame):
        B = pipe.B
        self.pipe, self.torch, self.B = pipe, torch, B
        R = self.max_rows = B * rows_per_frame

        def dev(shape, dt=torch.float32):
            return [torch.empty(shape, dtype=dt, device=device) for _ in range(2)]

        def host(shape, dt=torch.float32):
            return [torch.empty(shape, dtype=dt).pin_memory() for _ in range(2)]

        self.d_kp, self.d_ori, self.d_des, self.d_off = dev((R, 4)), dev((R,)), dev((R, 128)), dev((B + 2,), torch.int32)
        self.h_kp, self.h_ori, self.h_des, self.h_off = host((R, 4)), host((R,)), host((R, 128)), host((B + 2,), torch.int32)
        self.dstream = torch.cuda.Stream(device=device)
        self.ev_off = [torch.cuda.Event() for _ in range(2)]
        self.ev_done = [None, None]
        self.rows = [0, 0]
        self.bytes = 0
        self.pending = []                       # batches whose rows are packed but not yet on their way

    def after_step(self):
        """Call right after Pipeline.step(): sends the batch before the previous one on its way and packs
        the batch just enqueued."""
        k = self.pipe.step_no - 1
        p = k % 2
        if len(self.pending) >= 2:
            self.fetch()                        # batch k - 2: its staging arrays are the ones batch k packs into
        cs = self.pipe.mstream                  # the compute stream
        if self.ev_done[p] is not None:
            cs.wait_event(self.ev_done[p])      # the staging arrays are free again
        self.pipe.h[p].pack_results(self.B, self.d_kp[p].data_ptr(), self.d_ori[p].data_ptr(),
                                    self.d_des[p].data_ptr(), self.d_off[p].data_ptr(), self.max_rows)
        with self.torch.cuda.stream(cs):
            self.h_off[p].copy_(self.d_off[p], non_blocking=True)
        self.ev_off[p].record(cs)
        self.pending.append(k)

    def fetch(self):
        """Enqueues the download of the oldest packed batch."""
        if not self.pending:
            return
        p, B = self.pending.pop(0) % 2, self.B
        self.ev_off[p].synchronize()            # the batch's kernels have finished; the next one's are queued
        n, held = int(self.h_off[p][B]), int(self.h_off[p][B + 1])
        if held > n:
            raise RuntimeError("a batch holds %d feature rows, the download buffers %d" % (held, self.max_rows))
        with self.torch.cuda.stream(self.dstream):
            self.h_kp[p][:n].copy_(self.d_kp[p][:n], non_blocking=True)
            self.h_ori[p][:n].copy_(self.d_ori[p][:n], non_blocking=True)
            self.h_des[p][:n].copy_(self.d_des[p][:n], non_blocking=True)
        ev = self.torch.cuda.Event()
        ev.record(self.dstream)
        self.ev_done[p] = ev
        self.rows[p] = n
        self.bytes += n * (16 + 4 + 512) + (B + 2) * 4

    def finish(self):
        while self.pending:
            self.fetch()
        self.dstream.synchronize()

    def frame(self, parity, f):
        """(keypoints, orientations, descriptors) of frame f of the batch last downloaded with this parity."""
        lo, hi = int(self.h_off[parity][f]), int(self.h_off[parity][f + 1])
        return (self.h_kp[parity][lo:hi].numpy(), self.h_ori[parity][lo:hi].numpy(),
                self.h_des[parity][lo:hi].numpy())


# ------------------------------------------------------- checker-side pieces

def good_rule(d0):
    """The SIFT good-match rule of common/kpmatch.h:177-202, in double like the reference."""
    d0 = np.asarray(d0, np.float64)
    mn = min(100.0, float(d0.min())) if len(d0) else 100.0
    return d0 <= max(2.5 * mn, 0.35)


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
            _ = good_rule(dist[:, 0])
        prev = des
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": orc.lib().orc_get_threads(), "kind": "port",
            "sample": "%d frames %dx%d %s, %s, OpenMP restatement of the reference -f 0 path (oracle/)"
                      % (done, w, h, "gray" if frames.ndim == 3 else "BGR",
                         "extract + match vs previous frame" if match else "extraction only")}


def checker_frame(frames, f=1, match=True):
    """What the CPU checker gets for frame f >= 1 of the batch (matched against frame f - 1): per-octave
    counts, the features, the mask of good matches and the nearest database row of every query."""
    import oracle_lib as orc
    import object_detection_amd as oda
    n, h, w = frames.shape[:3]
    S = orc.SiftOracle(w, h, 1 if frames.ndim == 3 else frames.shape[3])
    S.run(frames[f - 1])
    des0 = S.all_features()[3]
    rc = S.run(frames[f])
    counts = [S.count(o) for o in range(3)]
    des1 = S.all_features()[3]
    out = {"status": rc, "counts": counts, "features": S.all_features()}
    if match and len(des0) and len(des1):
        idx, dist = orc.match_top2(oda.pack_sift_rows(des1), oda.pack_sift_rows(des0))
        out["keep"] = good_rule(dist[:, 0])
        out["idx"] = idx[:, 0]
    return out


def compare_frame(pipe, want, match=True, f=1):
    """Frame f of the batch extracted last against `want` (checker_frame). Returns (ok, details)."""
    p = pipe.last()
    status, counts, _ = pipe.h[p].read_counts(f)
    det = {"frame": f, "keypoints": counts, "keypoints_checker": want["counts"]}
    ok = counts == want["counts"] and status == want["status"]
    if match and "keep" in want:
        nq = sum(counts)
        summ = pipe.summ[p][f].cpu().numpy()
        good = pipe.good[p][f, :nq].cpu().numpy().astype(bool)
        idx = pipe.idx[p][f, :nq, 0].cpu().numpy()
        keep = want["keep"]
        same_rows = len(keep) == nq
        det.update(good_matches=int(summ[0]), good_matches_checker=int(keep.sum()))
        det["match_indices_equal"] = bool(same_rows and np.array_equal(good, keep) and
                                          np.array_equal(idx[keep], want["idx"][keep]))
        ok = ok and same_rows and int(summ[0]) == int(keep.sum()) and int(summ[1]) == nq
        ok = ok and det["match_indices_equal"]
    det["ok"] = bool(ok)
    return bool(ok), det


# ------------------------------------------------------------ matcher block

def c5_rows(torch, dev, base, rows_lo, rows_hi):
    """Rows [lo, hi) of the matcher block's database: row j = base row j % len(base), rotated by
    j // len(base) elements, so that a million distinct SIFT-like rows come from 65536 seeded ones."""
    nb = base.shape[0]
    parts = []
    j = rows_lo
    while j < rows_hi:
        r = j // nb
        end = min(rows_hi, (r + 1) * nb)
        parts.append(torch.roll(base[j - r * nb:end - r * nb], shifts=r, dims=1))
        j = end
    return torch.cat(parts) if parts else torch.empty((0, base.shape[1]), dtype=base.dtype, device=dev)


def matcher_block(args, oda, torch, dev, local_rank, rank, world, dist, backend):
    """BASELINE.json configs[4]: `--match-rows` queries x as many database rows x 128, rows resident in
    HBM; a pass resolves every query against the whole database. Exact fp32 search and bf16-screened
    search (same bits) are timed one after the other. With several ranks the database rows are sharded
    over them and the library gathers the per-shard candidates over RCCL (mif_match_sharded_*)."""
    from object_detection_amd.distributed import shard_range
    from synth import sift_like_descriptors
    n, dim = args.match_rows, 128
    base = torch.from_numpy(sift_like_descriptors(min(n, 65536), seed=7, dim=dim)).to(dev)
    lo, hi = shard_range(n, rank, world)
    db = c5_rows(torch, dev, base, lo, hi)
    g = torch.Generator(device="cpu")
    g.manual_seed(7)
    perm = torch.randperm(n, generator=g).to(dev)           # query i is a noisy copy of database row perm[i]
    nb = base.shape[0]
    cols = (torch.arange(dim, device=dev)[None, :] - (perm // nb)[:, None]) % dim
    q = torch.gather(base[perm % nb], 1, cols)
    del cols
    gd = torch.Generator(device=dev)
    gd.manual_seed(11)
    q = torch.clamp(q + 8.0 * torch.randn(q.shape, generator=gd, device=dev), 0.0, 255.0).contiguous()
    idx = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    dst = torch.zeros((n, 2), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    rows = hi - lo
    if world == 1:
        m = oda.Matcher(n, max(rows, 1), device=local_rank)
        pair = [dict(q=q.data_ptr(), db=db.data_ptr(), idx2=idx.data_ptr(), dist2=dst.data_ptr())]

        def search():
            m.top2_batch_device(pair, n, rows, dim, dim, 0, stream=stream)
        how = "one GPU holds all rows"
        sharded = None
    else:
        m = oda.ShardedMatcher.from_process_group(dist, local_rank, max_query=n, max_shard_rows=max(rows, 1))
        sharded = m.ranks()                     # the library's own view: ranks in the communicator, RCCL or not

        def search():
            m.search(q.data_ptr(), n, db.data_ptr(), rows, lo, dim, dim, idx.data_ptr(), dst.data_ptr(),
                     stream=stream)
        how = ("database rows sharded over %d GPUs, top-2 candidates all-gathered over RCCL by the library "
               "(mif_match_sharded_search)" % world)

    def timed(screening):
        m.set_screening(screening)
        search()                                        # warm-up
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.match_steps):
            search()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = (time.perf_counter() - t0) / args.match_steps
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), idx.clone(), dst.clone()

    dt_exact, idx_e, dst_e = timed(False)
    dt_scr, idx_s, dst_s = timed(True)
    passes = m.last_screen_passes()         # 2, or 1 + 1/n when the first pass samples the rows
    flops = 2.0 * dim * float(n) * float(n)
    same = bool(torch.equal(idx_e, idx_s) and torch.equal(dst_e.view(torch.int32), dst_s.view(torch.int32)))
    hit = float((idx_s[:, 0].to(torch.int64) == perm).float().mean().item())
    return {
        "workload": "%d queries x %d database rows x %d, fp32 rows resident in HBM (BASELINE.json configs[4]); %s"
                    % (n, n, dim, how),
        "metric": "descriptor matches/sec", "unit": "matches/s", "n_gpus": world,
        "scaling": "strong" if world > 1 else None,
        "sharded": sharded,
        "value": n / dt_scr, "ms_per_pass": dt_scr * 1e3,
        "value_exact_fp32": n / dt_exact, "ms_per_pass_exact_fp32": dt_exact * 1e3,
        "screened_equals_exact": same, "true_neighbour_recovered": hit,
        "roofline_exact_fp32": {"bound": "mfma", "achieved": flops / dt_exact / 1e12,
                                "peak": MFMA_F32_PEAK_TFLOPS * world, "unit": "TFLOP/s",
                                "frac": flops / dt_exact / 1e12 / (MFMA_F32_PEAK_TFLOPS * world)},
        # the screened search runs the contraction `passes` times in bf16 (the first pass may look at a
        # fraction of the rows) and ranks a few rows per query in fp32
        "roofline_screened_bf16": {"bound": "mfma", "achieved": passes * flops / dt_scr / 1e12,
                                   "peak": MFMA_BF16_PEAK_TFLOPS * world, "unit": "TFLOP/s",
                                   "frac": passes * flops / dt_scr / 1e12 / (MFMA_BF16_PEAK_TFLOPS * world),
                                   "bf16_passes": passes},
    }, same


# ---------------------------------------------------------------- native leg

def run_native(args, oda, torch, frames, local_rank, world, rank, dist):
    """The timed loop over mif_video_submit_device / mif_video_collect (same schedule, made inside the
    library); prints a reduced line (no roofline: the stage events belong to the handles inside)."""
    vp = oda.VideoPipeline(args.width, args.height, args.channels, batch=args.batch, device=local_rank)
    if args.host_frames:
        # three pinned buffers in rotation, as the contract of mif_video_submit_host asks
        pinned = [frames.cpu().pin_memory() for _ in range(3)]
        turn = [0]

        def submit():
            vp.submit_host_ptr(pinned[turn[0] % 3].data_ptr(), args.batch)
            turn[0] += 1
    else:
        def submit():
            vp.submit_device(frames.data_ptr(), args.batch)
    for _ in range(args.warmup):
        submit()
    vp.collect()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        submit()
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
                          "config": {"workload": "native pipeline object (mif_video_*), same schedule" +
                                                 (", frames uploaded from pinned host memory" if args.host_frames else ""),
                                     "frames_per_step": args.batch,
                                     "keypoints_last_frame": list(res[-1].keypoints),
                                     "good_matches_last_frame": res[-1].good_matches}}))


# ------------------------------------------------- the other configurations

def demo_pair_leg(kind, repeats):
    """BASELINE.json configs[0] / configs[1]: box.pgm against scene.pgm through the C++ demo of the
    reference's command line (`-m object scene`: extract both, match, good-match filter, homography),
    run as a child process. `-r N` makes the demo repeat the sequence N times with its pipelines built and
    report the mean host-to-host milliseconds of every step. Returns the parsed report (checked against
    the CPU checker by check_pair_leg, after the GPU work)."""
    import subprocess
    import tempfile
    exe = os.path.join(DEMO_DIR, "demo_coopcl_sift" if kind == "sift" else "demo_coopcl_Surf")
    if not os.path.exists(exe):
        return {"error": "%s is missing: run `make -C object_detection_amd/cpp`" % exe}
    box, scene = os.path.join(GOLDEN, "box.pgm"), os.path.join(GOLDEN, "scene.pgm")
    with tempfile.TemporaryDirectory() as tmp:
        rep = os.path.join(tmp, "report.json")
        try:
            out = subprocess.run([exe, "-m", box, scene, "-f", "1", "-o", rep, "-r", str(repeats)],
                                 capture_output=True, text=True, timeout=180, cwd=tmp)
        except subprocess.TimeoutExpired:
            return {"error": "the demo did not finish within 180 s"}
        if out.returncode != 0 or not os.path.exists(rep):
            return {"error": "demo exit code %d: %s" % (out.returncode, (out.stderr or out.stdout)[-400:])}
        with open(rep) as fh:
            return json.load(fh)


def check_pair_leg(kind, report, budget_s=4.0):
    """The demo's report against the CPU checker run on the same two images, and the checker's own time for
    the same steps (extraction of both images + matching; the homography is host code either way) as the
    CPU restatement beside it."""
    import oracle_lib as orc
    import object_detection_amd as oda
    if "error" in report:
        return {"error": report["error"], "verified": False}
    box, scene = orc.read_pnm(os.path.join(GOLDEN, "box.pgm")), orc.read_pnm(os.path.join(GOLDEN, "scene.pgm"))

    def run_checker():
        if kind == "sift":
            feats = []
            for img in (box, scene):
                S = orc.SiftOracle(img.shape[1], img.shape[0])
                S.run(img)
                feats.append(S.all_features()[3])
            idx, dist = orc.match_top2(oda.pack_sift_rows(feats[0]), oda.pack_sift_rows(feats[1]))
            keep = good_rule(dist[:, 0])
        else:
            feats = []
            for img in (box, scene):
                S = orc.SurfOracle(img.shape[1], img.shape[0])
                S.run(img)
                feats.append(S.points()[:, 6:])
            idx, dist = orc.match_top2(feats[0], feats[1])
            keep = dist[:, 0] < np.float32(0.75) * dist[:, 1]
        return feats, idx, keep

    feats, idx, keep = run_checker()                          # warm (library load, first touch)
    done, t0 = 0, time.perf_counter()
    while done < 3 or (time.perf_counter() - t0 < budget_s and done < 200):
        run_checker()
        done += 1
    cpu_ms = (time.perf_counter() - t0) / done * 1e3
    tm = report.get("timing_ms", {})
    want = np.stack([np.nonzero(keep)[0], idx[keep, 0]], 1).tolist()
    got = [g[:2] for g in report["good_matches"]]
    ok = got == want and tm.get("feats_object") == len(feats[0]) and tm.get("feats_scene") == len(feats[1])
    ok = ok and bool(report["has_homography"]) and report["inliers"] >= 4
    gpu_ms = tm.get("total")
    return {
        "workload": "%s box.pgm (324x223) vs scene.pgm (512x384): extract both, match, good-match filter, "
                    "RANSAC homography; pixels in host memory -> results in host memory, through "
                    "demo_coopcl_%s -m" % (kind.upper(), "sift" if kind == "sift" else "Surf"),
        "ms_per_pair": gpu_ms, "ms_per_pair_min": tm.get("total_min"), "repeats": tm.get("repeats"),
        "ms_steps": {k: tm.get(k) for k in ("extract_object", "extract_scene", "convert", "match", "homography")},
        "keypoints": [tm.get("feats_object"), tm.get("feats_scene")],
        "good_matches": len(got), "inliers": report["inliers"],
        "scene_corners": report["scene_corners"],
        "verified": bool(ok),
        "cpu_baseline": {"value": cpu_ms, "unit": "ms per pair (extraction of both images + matching)",
                         "cores": orc.lib().orc_get_threads(), "kind": "port",
                         "sample": "%d repeats of the same pair through the OpenMP restatement (oracle/); the "
                                   "homography (%.3f ms) is the same host code on both sides and is not "
                                   "included" % (done, tm.get("homography") or 0.0)},
        "gpu_over_cpu": (cpu_ms / (gpu_ms - (tm.get("homography") or 0.0))) if gpu_ms else None,
    }


def surf_batch_leg(args, oda, torch, device, local_rank, frames):
    """BASELINE.json configs[1] at throughput: a batch of 1080p BGR frames resident in HBM through
    mif_surf_enqueue_batch; frames/s, the integral image's achieved bandwidth against 7 B/px (3 read, 4 written), frame 1
    compared with the CPU checker, and the checker's own rate beside it."""
    import oracle_lib as orc
    progress("SURF batch leg starts")
    B, w, h = args.config_batch, 1920, 1080
    # `frames`: two rendered 1080p BGR frames, repeated to fill the batch
    px = torch.from_numpy(frames).to(device).repeat((B + 1) // 2, 1, 1, 1)[:B].contiguous()
    ex = oda.SurfExtractor(w, h, 3, device=local_rank, batch=B)
    for _ in range(2):
        ex.enqueue_batch(px.data_ptr(), B)
    ex.sync()
    t0 = time.perf_counter()
    for _ in range(args.config_steps):
        ex.enqueue_batch(px.data_ptr(), B)
    ex.sync()
    dt = (time.perf_counter() - t0) / args.config_steps
    ex.set_timing(True)
    stage = np.zeros(len(ex.STAGES))
    for _ in range(args.config_steps):
        ex.enqueue_batch(px.data_ptr(), B)
        stage += ex.stage_times()
    stage /= args.config_steps
    ex.set_timing(False)
    pts, raw = ex.copy_points(1)
    S = orc.SurfOracle(w, h)
    t0 = time.perf_counter()
    n = S.run(frames[1])
    cpu_s = time.perf_counter() - t0
    P = S.points()
    ok = len(pts) == n and raw == S.raw_count()
    ok = ok and np.array_equal(ex.integral(1).view(np.uint32), S.integral().view(np.uint32))
    if ok and n:
        ok = bool(np.array_equal(pts["descriptor"].view(np.uint32), P[:, 6:].view(np.uint32)) and
                  np.array_equal(pts["orientation"].view(np.uint32), P[:, 3].view(np.uint32)) and
                  np.array_equal(pts["x"].view(np.uint32), P[:, 0].view(np.uint32)))
    done, t0 = 1, time.perf_counter()
    while time.perf_counter() - t0 < 3.0 and done < 40:
        S.run(frames[done % 2])
        done += 1
    cpu_fps = (done - 1) / (time.perf_counter() - t0) if done > 1 else 1.0 / cpu_s
    integ = SURF_INTEGRAL_BYTES_PER_PX * w * h * B / (stage[0] * 1e-3) / 1e9
    # memory-side bytes of the two scan kernels from the counter passes (they scale with the frame count)
    traffic = None
    try:
        with open(os.path.join(ROOT, SURF_TRAFFIC_FILE)) as fh:
            traffic = json.load(fh)["integral_bytes_per_frame"] * B
    except (OSError, KeyError, ValueError):
        pass
    return {
        "workload": "SURF extraction of %d synthetic 1920x1080 BGR frames per step, resident in HBM "
                    "(mif_surf_enqueue_batch)" % B,
        "value": B / dt, "unit": "frames/s", "ms_per_step": dt * 1e3, "steps": args.config_steps,
        "points_frame1": int(n),
        "stage_ms_per_batch": {k: float(v) for k, v in zip(ex.STAGES, stage)},
        "integral_fallbacks": ex.integral_fallbacks(),     # batches the launch-per-pass form had to redo: 0
        "roofline": {"bound": "hbm", "kernel": "integral image (k_integral_fused: row scan and column scan in one "
                                               "launch; the stage's time includes clearing the response pools)",
                     "achieved": integ, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": integ / HBM_PEAK_GBS,
                     "alg_bytes_per_px": SURF_INTEGRAL_BYTES_PER_PX,
                     "alg_unfused_bytes_per_px": SURF_INTEGRAL_UNFUSED_BYTES_PER_PX,
                     "achieved_alg_unfused": integ * SURF_INTEGRAL_UNFUSED_BYTES_PER_PX / SURF_INTEGRAL_BYTES_PER_PX,
                     "traffic": traffic,
                     "traffic_source": SURF_TRAFFIC_FILE + " (the integral-image launch, per step)"},
        "verified": bool(ok),
        "cpu_baseline": {"value": cpu_fps, "unit": "frames/s", "cores": orc.lib().orc_get_threads(), "kind": "port",
                         "sample": "%d frames 1920x1080 BGR through the OpenMP restatement of the SURF pipeline "
                                   "(oracle/)" % max(done - 1, 1)},
    }


def sift_4k_leg(args, oda, torch, device, local_rank, frames_16):
    """BASELINE.json configs[3] on one GPU: a batch of 4K single-channel frames resident in HBM, SIFT
    extraction only; frames/s, the scale-space launches against the HBM peak, frame 1 compared with the
    CPU checker and the checker's own rate beside it."""
    import oracle_lib as orc
    progress("4K batch leg starts")
    B, w, h = args.config_batch, 3840, 2160
    reps = (B + len(frames_16) - 1) // len(frames_16)
    px = torch.from_numpy(frames_16).to(device).repeat(reps, 1, 1)[:B].contiguous()
    ex = oda.SiftExtractor(w, h, 1, device=local_rank, batch=B)
    for _ in range(2):
        ex.enqueue(px.data_ptr(), B)
    ex.sync()
    t0 = time.perf_counter()
    for _ in range(args.config_steps):
        ex.enqueue(px.data_ptr(), B)
    ex.sync()
    dt = (time.perf_counter() - t0) / args.config_steps
    ex.set_timing(True)
    ex.enqueue(px.data_ptr(), B)
    stage = ex.stage_times()
    ex.set_timing(False)
    opx = [h_ * w_ for (w_, h_, _) in (ex.octave_info(o) for o in range(3))]
    blur_ms = float(stage[:, 1].sum())
    achieved = min_bytes_per_frame(opx, 1) * B / (blur_ms * 1e-3) / 1e9
    rc, feat = ex.copy_results(1)
    S = orc.SiftOracle(w, h, 1)
    t0 = time.perf_counter()
    rc_o = S.run(frames_16[1 % len(frames_16)])
    cpu_s = time.perf_counter() - t0
    kp, octs, ori, des = S.all_features()
    ok = rc == rc_o and feat.counts == [S.count(o) for o in range(3)]
    ok = ok and np.array_equal(feat.keypoints.view(np.uint32), kp.view(np.uint32))
    ok = ok and np.array_equal(feat.orientations.view(np.uint32), ori.view(np.uint32))
    ok = ok and np.array_equal(feat.descriptors.view(np.uint32), des.view(np.uint32))
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 3.0 and done < 40:
        S.run(frames_16[done % len(frames_16)])
        done += 1
    cpu_fps = done / (time.perf_counter() - t0) if done else 1.0 / cpu_s
    # memory-side bytes of the scale-space launches per step, from the counter passes (taken at batch 16)
    traffic = None
    try:
        with open(os.path.join(ROOT, BLUR_4K_TRAFFIC_FILE)) as fh:
            traffic = json.load(fh)["blur_bytes_per_frame"] * B
    except (OSError, KeyError, ValueError):
        pass
    return {
        "workload": "SIFT extraction (keypoints, orientations, descriptors) of %d synthetic 3840x2160 "
                    "single-channel frames per step, resident in HBM (%d distinct frames, seed 4321, repeated "
                    "to fill the batch)" % (B, len(frames_16)),
        "value": B / dt, "unit": "frames/s", "ms_per_step": dt * 1e3, "steps": args.config_steps,
        "keypoints_frame1": feat.counts,
        "stage_ms_per_batch": {"scale_space": blur_ms, "detector": float(stage[:, 2].sum()),
                               "orient_desc": float(stage[:, 3].sum())},
        "roofline": {"bound": "hbm", "kernel": "k_blur_level / k_blur_pair", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "alg_bytes_per_frame": min_bytes_per_frame(opx, 1), "traffic": traffic,
                     "traffic_source": BLUR_4K_TRAFFIC_FILE + " (all scale-space launches of a step)"},
        "verified": bool(ok),
        "cpu_baseline": {"value": cpu_fps, "unit": "frames/s", "cores": orc.lib().orc_get_threads(), "kind": "port",
                         "sample": "%d frames 3840x2160 gray, extraction only, OpenMP restatement (oracle/)"
                                   % max(done, 1)},
    }


# ---------------------------------------------------- guarded matcher block

def guarded_matcher_block(run_block, rank, world, timeout_s, out, emit=print, leave=os._exit):
    """Runs the matcher block of a multi-rank run under a watchdog. Returns (block, ok, failed).

    The block contains the library's RCCL collective: a rank that fails before it leaves the others
    waiting. So a failure on any rank must end the whole job with a non-zero exit code, rank 0 included:
      * the watchdog (world > 1) fires after `timeout_s`: rank 0 prints the line as it stands with an
        `error` field, then every rank leaves with code 4 through `leave` (no barrier, no clean-up: the
        other ranks may be stuck inside the collective);
      * the block raises: ok = False and failed = True; the caller prints the line, skips the final barrier
        and exits non-zero, which makes the launcher stop the other ranks."""
    import threading

    def give_up():
        if rank == 0 and out is not None:
            out["matcher"] = {"error": "the sharded search did not finish within %d s" % timeout_s}
            emit(json.dumps(out), flush=True)
        leave(4)

    dog = threading.Timer(timeout_s, give_up) if world > 1 else None
    if dog is not None:
        dog.daemon = True
        dog.start()
    try:
        blk, ok = run_block()
        failed = False
    except Exception as exc:
        if world == 1:
            raise
        blk, ok, failed = {"error": repr(exc)}, False, True
    finally:
        if dog is not None:
            dog.cancel()
    return blk, ok, failed


# ---------------------------------------------------------------------- main

def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_only:
        return run_launch_only(rank, world, local_rank)
    # everything that forks or starts a child does so before the GPU is initialised
    pair_legs = None
    if rank == 0 and not args.no_configs:
        pair_legs = {"C1_sift_box_scene": demo_pair_leg("sift", args.pair_repeats),
                     "C2_surf_box_scene": demo_pair_leg("surf", args.pair_repeats)}
    frames_np = make_frames(args.width, args.height, args.batch, rank, args.channels, world)
    frames_4k, frames_surf = None, None
    if rank == 0 and not args.no_configs:
        frames_surf = frames_np[:2] if frames_np.shape[1:] == (1080, 1920, 3) and len(frames_np) >= 2 else \
            np.stack([_frame((1920, 1080, 1234, t, 3)) for t in range(2)])
        # 16 distinct 4K frames (seed 4321 + frame), repeated to fill the batch: rendering 64 would take longer
        # than measuring them, and frames of an extraction-only batch are independent
        frames_4k = make_frames(3840, 2160, min(16, args.config_batch), 0, 1, world, seed=4321)
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
    rank_devices = [local_rank]
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()

    device = torch.device("cuda", local_rank)

    def max_over_ranks(dt):
        t = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    frames = torch.from_numpy(frames_np).to(device)
    if args.native_pipeline:
        return run_native(args, oda, torch, frames, local_rank, world, rank, dist)
    pipe = Pipeline(oda, torch, local_rank, args.width, args.height, args.batch, serial=args.serial,
                    channels=args.channels, match=not args.no_match)

    def step_resident():
        pipe.step(frames.data_ptr())

    host_state = {}

    def step_host():
        """The same step with the batch uploaded from pinned host memory on copy streams (beside the
        previous batch's work) and the match summaries downloaded to pinned host memory."""
        if not host_state:
            ncopy = max(1, min(int(os.environ.get("BENCH_COPY_STREAMS", "1")), args.batch))
            host_state.update(
                host=torch.from_numpy(frames_np).pin_memory(),
                dev_buf=[torch.empty_like(frames), frames],
                cstreams=[torch.cuda.Stream(device=device) for _ in range(ncopy)],
                cuts=[args.batch * i // ncopy for i in range(ncopy + 1)],
                host_summ=[torch.empty((args.batch, 2), dtype=torch.int32).pin_memory() for _ in range(2)],
                marks=[])
        hs = host_state
        k = pipe.step_no
        hk = pipe.h[k % 2]
        # buffer k % 2 was last read by the first octave's scale space of batch k - 2 on this handle.
        # One copy stream: a single stream reaches the link's rate (57.6 GB/s measured, tools/micro/h2d_bw.hip)
        for ci, cs in enumerate(hs["cstreams"]):
            hk.wait_input_read(cs.cuda_stream)
            with torch.cuda.stream(cs):
                hs["dev_buf"][k % 2][hs["cuts"][ci]:hs["cuts"][ci + 1]].copy_(
                    hs["host"][hs["cuts"][ci]:hs["cuts"][ci + 1]], non_blocking=True)
        for cs in hs["cstreams"]:
            pipe._wait(hk.stream(), cs.cuda_stream)
        pipe.step(hs["dev_buf"][k % 2].data_ptr())
        if not pipe.split:
            # end of this batch's feature phase on the one compute stream: the distance between two such
            # marks is a step of the filled pipeline
            mark = torch.cuda.Event(enable_timing=True)
            mark.record(pipe.mstream)
            hs["marks"].append(mark)
        if k >= 1 and pipe.match and not pipe.serial:
            # the matcher of batch k - 1 has just been enqueued: its summaries go back to the host behind it
            with torch.cuda.stream(pipe.mstream):
                hs["host_summ"][(k - 1) % 2].copy_(pipe.summ[(k - 1) % 2], non_blocking=True)

    def steady_ms(steps):
        """ms per step between the end of the first and of the last timed step of the leg that has just run
        (the first step's upload has nothing to hide behind; a long sequence pays it once)."""
        marks = host_state.get("marks", [])[-steps:]
        if steps < 3 or len(marks) < steps:
            return None
        return marks[0].elapsed_time(marks[-1]) / (steps - 1)

    own_seconds = {}

    def timed_region(step, warmup, steps, finish=None, name=None):
        """Barrier + synchronize on both sides of exactly `steps` steps; max over the ranks. `finish` runs
        inside the timed region after the last step (work a leg still owes for its last batch). The rank's
        own time up to its synchronize, before the closing barrier, is kept under `name`."""
        for _ in range(warmup):
            step()
        if finish is not None:
            finish()
        pipe.sync()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if finish is not None:
            finish()
        pipe.sync()
        torch.cuda.synchronize()
        if name:
            own_seconds[name] = time.perf_counter() - t0
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    step = step_host if args.host_frames else step_resident
    for _ in range(args.warmup):
        step()
    pipe.sync()
    for h in pipe.h:
        h.set_timing(True)
    dt_max = timed_region(step, 0, args.steps, name="headline")
    per_rank_ms = gather_per_rank(dist, own_seconds["headline"] / args.steps * 1e3)

    # stage times of the last two batches of the timed region (one per handle), measured with HIP events on
    # the handles' own streams; blur_level is the dominant bandwidth kernel (6 launches per octave)
    timed = [pipe.h[(pipe.last() - i) % len(pipe.h)] for i in range(min(len(pipe.h), args.steps))]
    stage = np.mean([h.stage_times() for h in timed], axis=0)             # ms [octave][stage]
    for h in pipe.h:
        h.set_timing(False)
    px = [h_ * w_ for (w_, h_, _) in (pipe.h[0].octave_info(o) for o in range(3))]
    blur_ms = float(stage[:, 1].sum())
    launches = timed[0].blur_launches()       # 18 with one launch per level; two levels may share one
    min_bytes = min_bytes_per_frame(px, args.channels) * args.batch
    unfused_bytes = ALG_UNFUSED_BYTES_PER_PX_OCTAVE * sum(px) * args.batch
    achieved = min_bytes / (blur_ms * 1e-3) / 1e9
    # HBM-side bytes of the same launches from the PMC passes (FETCH_SIZE and WRITE_SIZE, separate
    # rocprofv3 runs of this script, summarised by tools/pmc_traffic.py); they scale with the frame count
    traffic = None
    try:
        with open(os.path.join(ROOT, BLUR_TRAFFIC_FILE)) as fh:
            pmc = json.load(fh)
        if args.width == 1920 and args.height == 1080:
            traffic = pmc["blur_bytes_per_frame"] * args.batch / launches
    except (OSError, KeyError, ValueError):
        pass
    summ = pipe.summ[pipe.last()].cpu().numpy()
    counts = pipe.h[pipe.last()].read_counts(args.batch - 1)

    progress("resident-frames steps timed")
    # the same frames through the CPU checker: frame 1 of the last batch must agree with it
    # (frames 1, batch / 2 and batch - 1 of the headline run; the later legs re-check frame 1)
    verified, verify_detail, want, verified_frames = None, None, None, None
    if not args.no_verify and rank == 0 and args.batch >= 2:
        want = checker_frame(frames_np, 1, match=not args.no_match)
        verified, verify_detail = compare_frame(pipe, want, match=not args.no_match)
        verified_frames = {1: verified}
        for f in sorted({max(1, args.batch // 2), args.batch - 1} - {1}):
            ok_f, _ = compare_frame(pipe, checker_frame(frames_np, f, match=not args.no_match),
                                    match=not args.no_match, f=f)
            verified_frames[f] = ok_f
            verified = bool(verified and ok_f)

    progress("headline batch checked against the CPU checker")
    # PCIe-inclusive rate: the same step fed from pinned host memory, summaries back on the host
    host_leg = None
    if not args.no_host_leg and not args.host_frames and not args.no_match:
        dt_host = timed_region(step_host, 2, args.host_steps)
        frames_host = args.batch * args.host_steps * world
        host_leg = {"value": frames_host / dt_host, "unit": "frames/s", "steps": args.host_steps,
                    "ms_per_step": dt_host / args.host_steps * 1e3,
                    "upload_GBps_per_gpu": args.batch * frames_np[0].nbytes * args.host_steps / dt_host / 1e9,
                    "window": "pinned host frames -> upload -> extract + match -> match summaries in pinned "
                              "host memory"}
        st = steady_ms(args.host_steps)
        if st:
            host_leg["ms_per_step_filled_pipeline"] = st
            host_leg["value_filled_pipeline"] = args.batch * world / (st * 1e-3)
        if want is not None:
            ok_h, _ = compare_frame(pipe, want, match=True)
            host_leg["verified"] = ok_h
            verified = bool(verified and ok_h)

    # the same, and every batch's features (keypoints, orientations, descriptors) go back to pinned host
    # memory as well: the window of the reference's call, pixels in host memory -> features in host memory
    feature_leg = None
    if host_leg is not None and not args.no_feature_leg and not pipe.split and not pipe.serial:
        fd = FeatureDownload(pipe, torch, device, max(pipe.cap // 4, 1))

        def step_host_features():
            step_host()
            fd.after_step()

        fd.bytes = 0
        dt_feat = timed_region(step_host_features, 2, args.host_steps, finish=fd.finish)
        # bytes of the timed steps only: the warm-up's are counted up to its finish()
        feat_bytes = fd.bytes * args.host_steps / (args.host_steps + 2)
        feature_leg = {"value": args.batch * args.host_steps * world / dt_feat, "unit": "frames/s",
                       "steps": args.host_steps, "ms_per_step": dt_feat / args.host_steps * 1e3,
                       "upload_GBps_per_gpu": args.batch * frames_np[0].nbytes * args.host_steps / dt_feat / 1e9,
                       "download_GBps_per_gpu": feat_bytes / dt_feat / 1e9,
                       "feature_rows_per_batch": fd.rows[pipe.last()],
                       "download_MB_per_frame": fd.rows[pipe.last()] * 532 / args.batch / 1e6,
                       "window": "pinned host frames -> upload -> extract + match -> keypoints, orientations, "
                                 "descriptors (packed per batch, mif_sift_pack_results) and match summaries in "
                                 "pinned host memory"}
        st = steady_ms(args.host_steps)
        if st:
            feature_leg["ms_per_step_filled_pipeline"] = st
            feature_leg["value_filled_pipeline"] = args.batch * world / (st * 1e-3)
        if want is not None:
            ok_f, _ = compare_frame(pipe, want, match=True)
            kp_h, ori_h, des_h = fd.frame(pipe.last(), 1)
            kp_w, _, ori_w, des_w = want["features"]
            ok_f = bool(ok_f and np.array_equal(kp_h.view(np.uint32), kp_w.view(np.uint32)) and
                        np.array_equal(ori_h.view(np.uint32), ori_w.view(np.uint32)) and
                        np.array_equal(des_h.view(np.uint32), des_w.view(np.uint32)))
            feature_leg["verified"] = ok_f
            verified = bool(verified and ok_f)
        del fd

    progress("host-to-host legs done")
    # informational: the same resident-frames step with the two handles free-running on a stream each.
    # Not `value`: the kernels of two batches then overlap in time, so no per-kernel duration (the roofline
    # block, a rocprofv3 summary) says what a kernel achieves by itself any more.
    two_streams = None
    if not args.no_two_streams_leg and not args.host_frames and not args.no_match and not pipe.split \
            and not args.serial and rank_devices.count(local_rank) == 1:
        try:
            pipe.enable_two_streams()
            dt_two = timed_region(step_resident, 3, args.host_steps)
            two_streams = {"value": args.batch * args.host_steps * world / dt_two, "unit": "frames/s",
                           "steps": args.host_steps, "ms_per_step": dt_two / args.host_steps * 1e3,
                           "window": "frames resident in HBM; two handles on a stream each, a matcher each, "
                                     "batches side by side"}
            if want is not None:
                ok_t, _ = compare_frame(pipe, want, match=True)
                two_streams["verified"] = ok_t
                verified = bool(verified and ok_t)
        except RuntimeError as exc:                      # e.g. no memory for the second matcher
            two_streams = {"skipped": repr(exc)}

    # yardstick measured on this GPU: a 1 GiB device-to-device copy (bytes read + bytes written)
    src_t = torch.empty(1 << 28, dtype=torch.float32, device=device)
    dst_t = torch.empty_like(src_t)
    dst_t.copy_(src_t)
    torch.cuda.synchronize()
    tc = time.perf_counter()
    for _ in range(5):
        dst_t.copy_(src_t)
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * src_t.numel() * 4 / (time.perf_counter() - tc) / 1e9
    del src_t, dst_t

    keypoints_last, good_last = counts[1], int(summ[-1, 0])
    world_seen = dist.get_world_size() if dist is not None else 1
    rc = 0
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
                                   ("matcher of batch k-1 beside the feature phase of batch k"
                                    if os.environ.get("BENCH_MATCH_STREAM") else
                                    "one stream: pyramid(k), matcher(k-1), features(k)"),
                       "keypoints_last_frame": keypoints_last, "good_matches_last_frame": good_last,
                       "parallelism": "frames sharded over %d GPU(s), no collective" % world,
                       "world_size": world_seen,
                       "rank_devices": rank_devices, "backend": backend if dist is not None else None},
            "roofline": {
                "bound": "hbm", "kernel": "k_blur_level / k_blur_pair (base image + H + V fused, %d launches per batch)" % launches,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": (BLUR_TRAFFIC_FILE + " (separate rocprofv3 --pmc passes of this command)")
                                  if traffic else None,
                "avg_launch_us": blur_ms * 1e3 / launches,
                "alg_bytes_per_launch": min_bytes / launches,
                "alg_unfused_bytes_per_launch": unfused_bytes / launches,
                "achieved_alg_unfused": unfused_bytes / (blur_ms * 1e-3) / 1e9,
                "achieved_traffic": (traffic / (blur_ms * 1e-3 / launches) / 1e9) if traffic else None,
                "measured_copy_GBps": copy_gbs,
            },
            "stage_ms_per_batch": {"base": float(stage[:, 0].sum()), "scale_space": blur_ms,
                                   "detector": float(stage[:, 2].sum()), "orient_desc": float(stage[:, 3].sum())},
            "matcher": None,
            "value_host_to_host": host_leg,
            "value_host_to_host_features": feature_leg,
            "per_rank_ms_per_step": per_rank_ms,
            "value_two_streams": two_streams,
            "verified": verified,
            "verified_frames": verified_frames,
            "verify_detail": verify_detail,
        }
    else:
        out = None

    # From here on the headline figures exist. A block further down that stalls must not cost the run its
    # line: at the deadline rank 0 prints what it has and leaves.
    watchdog = None
    if rank == 0 and args.deadline > 0:
        import threading

        def give_up():
            out["truncated"] = ("the run reached its deadline of %d s: blocks after the frame pipeline are missing "
                                "or partial" % args.deadline)
            print(json.dumps(out), flush=True)
            print("bench.py: deadline reached, line printed as far as the run got", file=sys.stderr, flush=True)
            os._exit(4)

        watchdog = threading.Timer(max(1.0, args.deadline - (time.perf_counter() - _T0)), give_up)
        watchdog.daemon = True
        watchdog.start()

    progress("frame pipeline legs done")
    # The frame pipeline's memory is released before the other blocks. The pinned buffers go first: torch's
    # host allocator records an event on every stream that used them when they are freed, and the handles'
    # stream is gone once the pipeline is.
    pipe.sync()
    host_state.clear()
    pipe.close()
    del pipe, frames, step, timed
    torch.cuda.empty_cache()

    # The other half of the metric: descriptor matches/s of the big search. With several ranks the block
    # contains the library's own RCCL collective; a stall or a failure there ends the whole job with a
    # non-zero exit code on every rank (guarded_matcher_block), after rank 0 has printed the line.
    match_ok, match_failed = True, False
    if args.match_rows > 0 and not args.no_match:
        if world > 1 and backend != "nccl":
            blk = {"skipped": "rehearsal with ranks sharing a GPU: RCCL needs one GPU per rank"}
        else:
            blk, match_ok, match_failed = guarded_matcher_block(
                lambda: matcher_block(args, oda, torch, device, local_rank, rank, world, dist, backend),
                rank, world, args.match_timeout, out)
        if rank == 0:
            out["matcher"] = blk

    progress("matcher block done")
    # The other configurations of BASELINE.json (one GPU; the pair legs ran as child processes above)
    configs_ok = True
    if rank == 0 and world == 1 and not args.no_configs:
        cfg = {"C1_sift_box_scene": check_pair_leg("sift", pair_legs["C1_sift_box_scene"]),
               "C2_surf_box_scene": check_pair_leg("surf", pair_legs["C2_surf_box_scene"]),
               "C2_surf_1080p_batch": surf_batch_leg(args, oda, torch, device, local_rank, frames_surf),
               "C3_sift_1080p_sequence": "the headline fields of this line (value, roofline, cpu_baseline, verified)",
               "C4_sift_4k_batch": sift_4k_leg(args, oda, torch, device, local_rank, frames_4k),
               "C5_match_1Mx1M": "the `matcher` block of this line"}
        configs_ok = all(v.get("verified", False) for v in cfg.values() if isinstance(v, dict))
        out["configs"] = cfg
        progress("configuration legs done")

    rc = 0
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames_np, args.cpu_seconds, match=not args.no_match)
            progress("CPU baseline done")
        if watchdog is not None:
            watchdog.cancel()
        print(json.dumps(out), flush=True)
        if verified is False or not match_ok or not configs_ok:
            print("bench.py: results differ from the CPU checker or between the two matcher forms, or the "
                  "matcher block failed", file=sys.stderr)
            rc = 1
    if match_failed:
        # this rank's part of the sharded search failed: the other ranks may be inside the collective, so
        # no barrier; the non-zero exit makes the launcher stop them
        sys.stdout.flush()
        os._exit(5)
    barrier()
    if dist is not None:
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
