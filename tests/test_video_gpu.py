"""GPU parity of the frame-sequence pipeline (mif_video_*): per-frame keypoint counts and good matches
against the previous frame, through the C ABI, against the CPU checker."""
import ctypes as C

import numpy as np
import pytest

import object_detection_amd as oda
from object_detection_amd import _lib
import oracle_lib as orc
from synth import blob_image

pytestmark = pytest.mark.gpu

W, H = 256, 192


def _sequence(n, channels):
    frames = [blob_image(W, H, seed=77, n_blobs=140, channels=channels, shift=(3 * t, 2 * t)) for t in range(n)]
    frames[4] = np.full_like(frames[4], 128)             # a flat frame: status -1, nothing to match
    return frames


def _expected(frames, channels):
    """(status, counts, good matches, query rows) per frame from the CPU checker."""
    out, prev = [], None
    for t, img in enumerate(frames):
        S = orc.SiftOracle(W, H, channels)
        rc = S.run(img)
        if rc == 0:
            counts = [S.count(o) for o in range(3)]
            des = S.all_features()[3]
        else:
            counts, des = [0, 0, 0], np.zeros((0, 128), np.float32)
        if t == 0:
            good = -1
        elif len(des) == 0:
            good = 0
        else:
            _, d = orc.match_top2(oda.pack_sift_rows(des), oda.pack_sift_rows(prev))
            mn = min(100.0, float(d[:, 0].astype(np.float64).min()))
            good = int((d[:, 0].astype(np.float64) <= max(2.5 * mn, 0.35)).sum())
        out.append((rc, counts, good, len(des)))
        prev = des
    return out


@pytest.mark.parametrize("channels", [1, 3])
def test_host_frames_in_uneven_batches(channels):
    frames = _sequence(8, channels)
    want = _expected(frames, channels)
    vp = oda.VideoPipeline(W, H, channels, batch=3)
    got = []
    for lo, hi in ((0, 3), (3, 6), (6, 7)):              # full, full, partial batch
        vp.submit_host(np.stack(frames[lo:hi]))
    got += vp.collect()
    vp.submit_host(np.stack(frames[7:8]))                # goes on after a collect; frame 7 follows frame 6
    got += vp.collect()
    assert len(got) == 8
    for t, (g, w) in enumerate(zip(got, want)):
        assert (g.status, list(g.keypoints), g.good_matches, g.query_rows) == w, t
    assert vp.collect() == []
    # extraction only: same counts, no match figures; matching resumes afterwards against the last frame
    vp.set_matching(False)
    vp.submit_host(np.stack(frames[0:3]))
    off = vp.collect()
    assert [(g.status, list(g.keypoints), g.query_rows) for g in off] == [(w[0], w[1], w[3]) for w in want[0:3]]
    assert [g.good_matches for g in off] == [-1, -1, -1]
    vp.set_matching(True)
    vp.submit_host(np.stack(frames[3:4]))
    on = vp.collect()
    assert (on[0].status, list(on[0].keypoints), on[0].good_matches, on[0].query_rows) == want[3]
    vp.close()


def test_device_frames_and_argument_errors():
    import torch
    frames = _sequence(6, 3)
    want = _expected(frames, 3)
    dev = torch.device("cuda:0")
    vp = oda.VideoPipeline(W, H, 3, batch=4)
    a = torch.from_numpy(np.stack(frames[0:4])).to(dev)
    b = torch.from_numpy(np.stack(frames[4:6])).to(dev)
    torch.cuda.synchronize()
    vp.submit_device(a.data_ptr(), 4)
    vp.submit_device(b.data_ptr(), 2)
    got = vp.collect()
    assert [(g.status, list(g.keypoints), g.good_matches, g.query_rows) for g in got] == want
    lib = oda.load_library()
    assert lib.mif_video_submit_device(vp._h, C.c_void_p(a.data_ptr()), 5) == _lib.MIF_ERR_CAPACITY
    assert lib.mif_video_submit_device(vp._h, None, 1) == _lib.MIF_ERR_INVALID_ARGUMENT
    vp.submit_device(a.data_ptr(), 4)
    n = C.c_int()
    small = (_lib.VideoFrame * 2)()
    assert lib.mif_video_collect(vp._h, small, 2, C.byref(n)) == _lib.MIF_ERR_CAPACITY and n.value == 0
    rest = vp.collect()                                  # nothing was consumed by the failed call
    assert len(rest) == 4 and rest[1].good_matches == want[1][2]
    vp.close()


def _rank_main(rank, world, port, out_dir):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import object_detection_amd as oda_
    from object_detection_amd.distributed import sharded_sequence
    frames = _sequence(9, 3)

    def process(lo, hi):
        vp = oda_.VideoPipeline(W, H, 3, batch=4)
        for a in range(lo, hi, 4):
            vp.submit_host(np.stack(frames[a:min(a + 4, hi)]))
        recs = vp.collect()
        vp.close()
        return [tuple(r) for r in recs]

    got = sharded_sequence(len(frames), process)
    if rank == 0:
        single = process(0, len(frames))
        with open(os.path.join(out_dir, "ok"), "w") as fh:
            fh.write("1" if got == single and len(got) == 9 else "0 %r %r" % (got, single))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_a_sequence(tmp_path):
    """Two processes (gloo; they share this box's one GPU) take one block of the sequence each, the second
    block extended by its predecessor frame: the gathered records equal a single run's."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.start_processes(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    assert open(tmp_path / "ok").read() == "1"


def test_full_hd_bgr_sequence_against_checker():
    """Four 1920x1080 BGR frames (the geometry of the headline configuration) through mif_video_* in
    batches of 3 + 1: status, keypoints per octave, descriptor rows and good matches against the previous
    frame equal the CPU checker's."""
    w, h = 1920, 1080
    frames = [blob_image(w, h, seed=1234, channels=3, shift=(2 * t, t), wrap=True) for t in range(4)]
    want, prev = [], None
    S = orc.SiftOracle(w, h, 3)
    for t, img in enumerate(frames):
        rc = S.run(img)
        assert rc == 0
        counts = [S.count(o) for o in range(3)]
        des = S.all_features()[3]
        good = -1
        if t > 0:
            _, d = orc.match_top2(oda.pack_sift_rows(des), oda.pack_sift_rows(prev))
            d0 = d[:, 0].astype(np.float64)
            good = int((d0 <= max(2.5 * min(100.0, float(d0.min())), 0.35)).sum())
        want.append((rc, counts, good, len(des)))
        prev = des
    vp = oda.VideoPipeline(w, h, 3, batch=3)
    vp.submit_host(np.stack(frames[0:3]))
    vp.submit_host(np.stack(frames[3:4]))
    got = vp.collect()
    vp.close()
    assert [(g.status, list(g.keypoints), g.good_matches, g.query_rows) for g in got] == want
    assert all(w_[2] > 20 for w_ in want[1:]) and all(sum(w_[1]) > 2000 for w_ in want)


def test_pinned_host_buffers_reused_as_documented():
    """mif_video_submit_host with pinned memory copies asynchronously. Three pinned buffers used
    in rotation and overwritten as soon as the contract allows (once the call after next has returned)
    must give the records of the same frames submitted from fresh buffers."""
    import torch
    n_batches, B = 6, 4
    frames = [blob_image(W, H, seed=91, n_blobs=140, channels=3, shift=(3 * t, 2 * t)) for t in range(n_batches * B)]
    vp = oda.VideoPipeline(W, H, 3, batch=B)
    for k in range(n_batches):
        vp.submit_host(np.stack(frames[k * B:(k + 1) * B]))
    want = [tuple(r) for r in vp.collect()]
    vp.close()
    vp = oda.VideoPipeline(W, H, 3, batch=B)
    pinned = [torch.empty((B, H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(3)]
    for k in range(n_batches):
        buf = pinned[k % 3]                     # buffer k % 3 was handed over by call k - 3: calls k - 2 and k - 1 have returned
        buf.copy_(torch.from_numpy(np.stack(frames[k * B:(k + 1) * B])))
        vp.submit_host_ptr(buf.data_ptr(), B)
        if k >= 2:
            pinned[(k - 2) % 3].fill_(0)        # ... so the buffer of call k - 2 may be overwritten now
    got = [tuple(r) for r in vp.collect()]
    vp.close()
    assert got == want and len(got) == n_batches * B
