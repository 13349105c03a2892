"""N > 1 host logic rehearsed on CPU: two gloo ranks shard a frame sequence and a descriptor database;
the per-shard work is done by the CPU checker so that only the distributed plumbing is under test."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as orc
    from object_detection_amd.distributed import ShardedDatabaseMatcher, shard_range
    from synth import blob_image, sift_like_descriptors
    orc.lib().orc_set_threads(2)

    # database sharding: 1000 rows (not divisible by the world size on purpose with world = 3)
    db = sift_like_descriptors(1001, 3)
    q = sift_like_descriptors(157, 4)
    db[700] = q[5]
    db[10] = db[990] = q[9]          # a tie across two shards -> the lower global index wins
    lo, hi = shard_range(len(db), rank, world)
    sm = ShardedDatabaseMatcher(hi - lo, lo, local_top2=lambda a, b: orc.match_top2(a, b, return_scores=True))
    idx, d = sm.match(q, db[lo:hi])
    ref_idx, ref_d = orc.match_top2(q, db)
    ok_match = bool(np.array_equal(idx, ref_idx) and np.array_equal(d.view(np.uint32), ref_d.view(np.uint32)))

    # frame sharding: every rank extracts its block of a 6-frame sequence; counts are gathered for the check only
    n_frames = 6
    flo, fhi = shard_range(n_frames, rank, world)
    counts = torch.zeros(n_frames, 3, dtype=torch.int64)
    S = orc.SiftOracle(160, 120)
    for t in range(flo, fhi):
        S.run(blob_image(160, 120, seed=9, shift=(2 * t, t)), want_descriptors=False)
        counts[t] = torch.tensor([S.count(o) for o in range(3)])
    dist.all_reduce(counts)

    # a sequence whose records depend on the previous frame: the shards must reproduce the single run
    from object_detection_amd.distributed import sharded_sequence
    calls = []

    def process(lo, hi):
        calls.append((lo, hi))
        return [(t, -1 if t == lo else 1000 * t + (t - 1)) for t in range(lo, hi)]   # (frame, "match vs previous")

    seq = sharded_sequence(7, process)
    want = [(0, -1)] + [(t, 1000 * t + t - 1) for t in range(1, 7)]
    ok_seq = seq == want and len(calls) == 1 and calls[0][0] == max(shard_range(7, rank, world)[0] - 1, 0)
    ok_match = ok_match and ok_seq
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([int(ok_match)] + counts.flatten().tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_gloo_sharding(world, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    import oracle_lib as orc
    from synth import blob_image
    S = orc.SiftOracle(160, 120)
    want = []
    for t in range(6):
        S.run(blob_image(160, 120, seed=9, shift=(2 * t, t)), want_descriptors=False)
        want += [S.count(o) for o in range(3)]
    for r in range(world):
        got = np.load(tmp_path / ("r%d.npy" % r))
        assert got[0] == 1, "rank %d: sharded match differs from the single search" % r
        assert got[1:].tolist() == want


def test_shard_range_and_merge_unit():
    from object_detection_amd.distributed import merge_top2, shard_range
    for n in (0, 1, 7, 16):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b[1] - b[0] for b in blocks]
            assert max(sizes) - min(sizes) <= 1
    inf = float("inf")
    scores = torch.tensor([[[1.0, 5.0]], [[1.0, 2.0]], [[inf, inf]]])        # 3 shards, 1 query
    idx = torch.tensor([[[40, 41]], [[7, 90]], [[-1, -1]]], dtype=torch.int32)
    dists = torch.tensor([[[1.0, 2.2]], [[1.0, 1.4]], [[inf, inf]]])
    wi, wd = merge_top2(scores, idx, dists)
    assert wi.tolist() == [[7, 40]] and wd.tolist() == [[1.0, 1.0]]           # score tie -> lower index first
    scores = torch.tensor([[[3.0, inf]], [[inf, inf]]])
    idx = torch.tensor([[[2, -1]], [[-1, -1]]], dtype=torch.int32)
    dists = torch.tensor([[[1.7, inf]], [[inf, inf]]])
    wi, wd = merge_top2(scores, idx, dists)
    assert wi.tolist() == [[2, -1]] and wd[0, 0] == 1.7 and torch.isinf(wd[0, 1])


def test_bench_gpus_flag_starts_that_many_ranks():
    """`bench.py --gpus N` without a launcher around it starts N ranks itself (launch-only mode: the
    process group forms over gloo, no GPU work) and rank 0 reports the world it saw."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    for n in (1, 2, 3):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--launch-only"],
                             env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == n and rec["config"]["world_size"] == n
        assert rec["config"]["rank_devices"] == list(range(n))
        # the multi-rank fields of the full line: one time per rank, the sharded matcher's view of the
        # ranks, and a frame-rendering pool that all ranks of the node can afford together
        assert len(rec["per_rank_ms_per_step"]) == n and all(t >= 0 for t in rec["per_rank_ms_per_step"])
        assert rec["matcher"]["sharded"]["nranks"] == n and "uses_rccl" in rec["matcher"]["sharded"]
        assert 1 <= rec["frame_workers"] <= max(1, min(8, (os.cpu_count() or 1) // n))


def test_bench_launcher_reports_a_failing_rank():
    """A rank that dies makes the parent stop the others and exit non-zero."""
    import subprocess
    if torch.cuda.is_available():
        pytest.skip("needs a machine without GPUs: there the ranks of a real run stop at once")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    # without a GPU every rank of a real run stops with "bench.py needs a GPU"; tiny frames keep it quick
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2",
                          "--width", "64", "--height", "48", "--no-configs"], env=env, capture_output=True, text=True, timeout=180)
    assert out.returncode != 0
    assert "needs a GPU" in out.stderr


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_matcher_block_failures_are_failures():
    """A multi-rank matcher block that raises is reported as failed (not as a pass with an `error` field),
    a single-rank one re-raises, and a block that stalls makes the watchdog print the line with the error
    and leave with a non-zero code on every rank, rank 0 included."""
    import json
    import threading
    bench = _load_bench()

    def boom():
        raise RuntimeError("ncclAllGather: unhandled system error")

    blk, ok, failed = bench.guarded_matcher_block(boom, rank=0, world=2, timeout_s=60, out={})
    assert "unhandled system error" in blk["error"] and ok is False and failed is True
    with pytest.raises(RuntimeError):
        bench.guarded_matcher_block(boom, rank=0, world=1, timeout_s=60, out={})
    blk, ok, failed = bench.guarded_matcher_block(lambda: ({"value": 1.0}, True), rank=1, world=2, timeout_s=60, out=None)
    assert blk == {"value": 1.0} and ok is True and failed is False

    for rank in (0, 1):
        printed, left, release = [], [], threading.Event()

        def emit(text, flush=False):
            printed.append(text)

        def leave(code):
            left.append(code)
            release.set()

        def stalls():
            release.wait(30)                   # a collective the other ranks never join
            return {"value": 0.0}, True

        out = {"metric": "x", "matcher": None}
        bench.guarded_matcher_block(stalls, rank=rank, world=2, timeout_s=0.2, out=out, emit=emit, leave=leave)
        assert left == [4]                     # non-zero on every rank
        if rank == 0:
            assert len(printed) == 1 and "did not finish" in json.loads(printed[0])["matcher"]["error"]
        else:
            assert printed == []
