"""GPU parity of the batched, device-resident path (what bench.py runs): a batch of frames through
mif_sift_enqueue, then frame-to-previous-frame matching with the SIFT good-match filter on the device."""
import os

import numpy as np
import pytest

import object_detection_amd as oda
from object_detection_amd import _lib
import oracle_lib as orc
from synth import blob_image, dense_image

pytestmark = pytest.mark.gpu


def _frames():
    return [blob_image(256, 192, seed=1), blob_image(256, 192, seed=1, shift=(3, 2)),
            dense_image(256, 192, 0), np.full((192, 256), 90, np.uint8), blob_image(256, 192, seed=9)]


def test_batched_extract_and_match_against_checker():
    import torch
    frames = _frames()
    B = len(frames)
    dev = torch.device("cuda:0")
    px = torch.from_numpy(np.stack(frames)).to(dev)
    ex = oda.SiftExtractor(256, 192, 1, batch=B)
    ex.set_timing(True)
    ex.enqueue(px.data_ptr(), B)
    kp_p, ori_p, des_p, cnt_p = ex.device_results()
    cap = ex.capacity
    oracle = []
    for f, img in enumerate(frames):
        S = orc.SiftOracle(256, 192)
        rc_o = S.run(img)
        rc_g, feat = ex.copy_results(f)
        assert rc_g == rc_o, f
        if rc_o == 0:
            kp, octs, ori, des = S.all_features()
            assert feat.counts == [S.count(o) for o in range(3)]
            assert np.array_equal(feat.keypoints.view(np.uint32), kp.view(np.uint32))
            assert np.array_equal(feat.orientations.view(np.uint32), ori.view(np.uint32))
            assert np.array_equal(feat.descriptors.view(np.uint32), des.view(np.uint32))
            oracle.append(des)
        else:
            assert feat.counts == [0, 0, 0]
            oracle.append(np.zeros((0, 128), np.float32))
    t = ex.stage_times()
    assert t.shape == (3, 4) and np.all(t >= 0) and t.sum() > 0

    # frame b against frame b-1 with device-side row counts, shifted rows (dim 127 of ld 128) and the SIFT filter
    idx = torch.full((B, cap, 2), -7, dtype=torch.int32, device=dev)
    dist = torch.zeros((B, cap, 2), dtype=torch.float32, device=dev)
    good = torch.zeros((B, cap), dtype=torch.uint8, device=dev)
    summ = torch.full((B, 2), -1, dtype=torch.int32, device=dev)
    mnmx = torch.zeros((B, 2), dtype=torch.float32, device=dev)
    m = oda.Matcher(cap, cap, max_pairs=B)
    pairs = []
    for b in range(1, B):
        pairs.append(dict(q=des_p + b * cap * 512, db=des_p + (b - 1) * cap * 512,
                          nq=cnt_p + b * 32 + 12, nd=cnt_p + (b - 1) * 32 + 12,
                          idx2=idx[b].data_ptr(), dist2=dist[b].data_ptr(), good=good[b].data_ptr(),
                          summary=summ[b].data_ptr(), minmax=mnmx[b].data_ptr()))
    m.top2_batch_device(pairs, cap, cap, 127, 128, _lib.FILTER_SIFT, stream=ex.stream())
    ex.sync()
    idx, dist, good, summ, mnmx = idx.cpu().numpy(), dist.cpu().numpy(), good.cpu().numpy(), summ.cpu().numpy(), mnmx.cpu().numpy()
    for b in range(1, B):
        q, db = oracle[b], oracle[b - 1]
        nq = len(q)
        assert summ[b, 1] == nq
        if nq == 0:
            assert summ[b, 0] == 0
            continue
        i_o, d_o = orc.match_top2(oda.pack_sift_rows(q), oda.pack_sift_rows(db))
        assert np.array_equal(idx[b, :nq], i_o)
        assert np.array_equal(dist[b, :nq].view(np.uint32), d_o.view(np.uint32))
        mn = min(100.0, float(d_o[:, 0].astype(np.float64).min()))
        keep = d_o[:, 0].astype(np.float64) <= max(2.5 * mn, 0.35)
        assert np.array_equal(good[b, :nq].astype(bool), keep)
        assert summ[b, 0] == int(keep.sum())
        assert mnmx[b, 0] == np.float32(mn)


def test_ratio_filter_on_device():
    import torch
    from synth import sift_like_descriptors
    dev = torch.device("cuda:0")
    a = sift_like_descriptors(300, 6)
    b = sift_like_descriptors(500, 7)
    rng = np.random.RandomState(8)
    b[:150] = np.clip(a[:150] + rng.normal(0, 4, (150, 128)), 0, 255).astype(np.float32)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    idx = torch.zeros((300, 2), dtype=torch.int32, device=dev)
    dist = torch.zeros((300, 2), dtype=torch.float32, device=dev)
    good = torch.zeros(300, dtype=torch.uint8, device=dev)
    summ = torch.zeros(2, dtype=torch.int32, device=dev)
    m = oda.Matcher(300, 500)
    m.top2_batch_device([dict(q=ta.data_ptr(), db=tb.data_ptr(), idx2=idx.data_ptr(), dist2=dist.data_ptr(),
                              good=good.data_ptr(), summary=summ.data_ptr())], 300, 500, 128, 128,
                        _lib.FILTER_RATIO)
    torch.cuda.synchronize()
    i_o, d_o = orc.match_top2(a, b)
    keep = d_o[:, 0] < np.float32(0.75) * d_o[:, 1]
    assert np.array_equal(idx.cpu().numpy(), i_o)
    assert np.array_equal(good.cpu().numpy().astype(bool), keep)
    assert summ.cpu().numpy().tolist() == [int(keep.sum()), 300]
    assert keep.sum() >= 140


def test_4k_gray_batch_parity():
    """3840x2160 single-channel frames (the 4K configuration of the benchmark list) through the
    batched device path: every frame of the batch equals the checker's run bit for bit."""
    import torch
    from synth import blob_image
    frames = [blob_image(3840, 2160, seed=4321 + t, n_blobs=6000) for t in range(2)]
    dev = torch.device("cuda:0")
    px = torch.from_numpy(np.stack(frames)).to(dev)
    ex = oda.SiftExtractor(3840, 2160, 1, batch=2)
    assert [ex.octave_info(o)[:2] for o in range(3)] == [(3840, 2160), (1920, 1080), (960, 540)]
    ex.enqueue(px.data_ptr(), 2)
    S = orc.SiftOracle(3840, 2160)
    for f, img in enumerate(frames):
        rc_o = S.run(img)
        rc_g, feat = ex.copy_results(f)
        assert rc_g == rc_o == 0
        kp, octs, ori, des = S.all_features()
        assert feat.counts == [S.count(o) for o in range(3)] and sum(feat.counts) > 1000
        assert np.array_equal(feat.keypoints.view(np.uint32), kp.view(np.uint32))
        assert np.array_equal(feat.orientations.view(np.uint32), ori.view(np.uint32))
        assert np.array_equal(feat.descriptors.view(np.uint32), des.view(np.uint32))


def test_two_phase_enqueue_snapshot_and_stream_order():
    """mif_sift_enqueue_pyramid + mif_sift_enqueue_features equal mif_sift_enqueue; the snapshot of a
    frame holds its descriptor rows and count; mif_stream_wait_stream orders a foreign stream."""
    import ctypes as C
    import torch
    frames = _frames()
    B = len(frames)
    dev = torch.device("cuda:0")
    px = torch.from_numpy(np.stack(frames)).to(dev)
    one = oda.SiftExtractor(256, 192, 1, batch=B)
    one.enqueue(px.data_ptr(), B)
    two = oda.SiftExtractor(256, 192, 1, batch=B)
    two.enqueue_pyramid(px.data_ptr(), B)
    two.enqueue_features(True)
    cap = two.capacity
    snap = torch.full((cap, 128), -1.0, dtype=torch.float32, device=dev)
    cnt = torch.full((1,), -1, dtype=torch.int32, device=dev)
    two.snapshot_frame(B - 1, snap.data_ptr(), cnt.data_ptr())
    side = torch.cuda.Stream(device=dev)
    lib = oda.load_library()
    _lib.check(lib.mif_stream_wait_stream(0, C.c_void_p(side.cuda_stream), C.c_void_p(two.stream())))
    with torch.cuda.stream(side):
        snap2 = snap.clone()            # ordered after the snapshot copy by the wait above
    side.synchronize()
    for f in range(B):
        rc1, a = one.copy_results(f)
        rc2, b = two.copy_results(f)
        assert rc1 == rc2 and a.counts == b.counts
        assert np.array_equal(a.keypoints.view(np.uint32), b.keypoints.view(np.uint32))
        assert np.array_equal(a.descriptors.view(np.uint32), b.descriptors.view(np.uint32))
    n = len(b.descriptors)
    assert int(cnt.cpu()[0]) == n and n > 0
    assert np.array_equal(snap2.cpu().numpy()[:n], b.descriptors)
    assert np.all(snap2.cpu().numpy()[n:] == -1.0)          # only the frame's rows are written
    # the snapshot moves 16-byte pieces: a misaligned buffer is refused before anything is enqueued
    with pytest.raises(_lib.MifError):
        two.snapshot_frame(B - 1, snap.data_ptr() + 4, cnt.data_ptr())
    # the feature phase needs a pending scale space
    with pytest.raises(_lib.MifError):
        two.enqueue_features(True)


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the agreed keys (small batch, bounded CPU sample)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--batch", "4", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["unit"] == "frames/s"
    assert "workload" in d["config"] and d["config"]["frames_per_step"] == 4
    assert abs(d["value"] - 4 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
    # the legs beside the headline (uploads from pinned memory; two free-running streams) ran and were checked
    assert d["verified"] is True
    for leg in ("value_host_to_host", "value_two_streams"):
        assert d[leg]["value"] > 0 and d[leg]["verified"] is True, leg


@pytest.mark.parametrize("geom", [(101, 67, 3, 3), (203, 131, 1, 2), (75, 49, 3, 4), (130, 51, 1, 3), (9, 40, 3, 2)])
def test_odd_sized_batches_against_checker(geom):
    """Frames of odd sizes back to back (a BGR frame then starts at an odd byte address and its pixels are
    fetched as unaligned words, the decimation uses its flat indexing): every frame of the batch equals
    the checker in counts, keypoints, orientations and descriptors."""
    import torch
    w, h, c, B = geom
    frames = [blob_image(w, h, seed=31 * w + h + i, channels=c, shift=(i, 2 * i)) for i in range(B)]
    px = torch.from_numpy(np.stack(frames)).to(torch.device("cuda:0"))
    ex = oda.SiftExtractor(w, h, c, batch=B)
    ex.enqueue(px.data_ptr(), B)
    for f, img in enumerate(frames):
        S = orc.SiftOracle(w, h, c)
        rc_o = S.run(img)
        rc_g, feat = ex.copy_results(f)
        assert rc_g == rc_o, f
        if rc_o != 0:
            continue
        kp, octs, ori, des = S.all_features()
        assert feat.counts == [S.count(o) for o in range(3)], f
        assert np.array_equal(feat.keypoints.view(np.uint32), kp.view(np.uint32)), f
        assert np.array_equal(feat.orientations.view(np.uint32), ori.view(np.uint32)), f
        assert np.array_equal(feat.descriptors.view(np.uint32), des.view(np.uint32)), f


def test_extraction_on_a_stream_with_a_share_of_the_compute_units():
    """mif_stream_create_masked: a handle enqueued on a stream that owns the low 64 of 256 compute units
    returns the bits of a handle on its own stream (the kernels' grids do not depend on where they run)."""
    import ctypes as C
    import torch
    frames = _frames()
    B = len(frames)
    px = torch.from_numpy(np.stack(frames)).to(torch.device("cuda:0"))
    lib = oda.load_library()
    mask = (C.c_uint32 * 8)(0xFFFFFFFF, 0xFFFFFFFF, 0, 0, 0, 0, 0, 0)
    st = C.c_void_p()
    _lib.check(lib.mif_stream_create_masked(0, mask, 8, C.byref(st)))
    try:
        one = oda.SiftExtractor(256, 192, 1, batch=B)
        one.enqueue(px.data_ptr(), B)
        two = oda.SiftExtractor(256, 192, 1, batch=B)
        two.set_stream(st.value)
        two.enqueue(px.data_ptr(), B)
        total = 0
        for f in range(B):
            rc1, a = one.copy_results(f)
            rc2, b = two.copy_results(f)
            assert rc1 == rc2 and a.counts == b.counts
            assert np.array_equal(a.keypoints.view(np.uint32), b.keypoints.view(np.uint32))
            assert np.array_equal(a.descriptors.view(np.uint32), b.descriptors.view(np.uint32))
            total += sum(a.counts)
        assert total > 100
        two.set_stream(None)
    finally:
        _lib.check(lib.mif_stream_destroy(0, st))
    assert lib.mif_stream_create_masked(0, None, 8, C.byref(st)) != 0
    sparse = (C.c_uint32 * 8)(0x55555555, 0, 0, 0, 0, 0, 0, 0)
    assert lib.mif_stream_create_masked(0, sparse, 8, C.byref(st)) != 0 and b"contiguous" in lib.mif_last_error()
    # refused before the runtime is called: a run that leaves some XCD without a compute unit, an empty
    # mask, and bits beyond the device's compute units
    few = (C.c_uint32 * 8)(0xF, 0, 0, 0, 0, 0, 0, 0)
    assert lib.mif_stream_create_masked(0, few, 8, C.byref(st)) != 0 and b"XCDs" in lib.mif_last_error()
    none = (C.c_uint32 * 8)()
    assert lib.mif_stream_create_masked(0, none, 8, C.byref(st)) != 0
    beyond = (C.c_uint32 * 16)(*([0] * 7 + [0xFF000000, 0xFF] + [0] * 7))
    assert lib.mif_stream_create_masked(0, beyond, 16, C.byref(st)) != 0 and b"compute units" in lib.mif_last_error()


def test_pack_results_equals_per_frame_copies():
    """mif_sift_pack_results: the batch's rows back to back (frame-major, octave-major inside a frame, as
    sift.cpp:1889-1911 appends them) equal the per-frame copies, which the tests above compare with the
    checker; a frame without keypoints in some octave contributes nothing; max_rows cuts from the end."""
    import torch
    frames = _frames()
    B = len(frames)
    dev = torch.device("cuda:0")
    px = torch.from_numpy(np.stack(frames)).to(dev)
    ex = oda.SiftExtractor(256, 192, 1, batch=B)
    ex.enqueue(px.data_ptr(), B)
    per_frame = [ex.copy_results(f)[1] for f in range(B)]
    totals = [len(f.keypoints) for f in per_frame]
    assert sum(totals) > 100 and 0 in totals            # the flat frame has no keypoints
    for nframes, max_rows in ((B, B * ex.capacity), (3, B * ex.capacity), (B, sum(totals) - 5), (B, 0)):
        kp = torch.full((max(max_rows, 1), 4), -3.0, device=dev)
        ori = torch.full((max(max_rows, 1),), -3.0, device=dev)
        des = torch.full((max(max_rows, 1), 128), -3.0, device=dev)
        off = torch.full((nframes + 2,), -9, dtype=torch.int32, device=dev)
        ex.pack_results(nframes, kp.data_ptr(), ori.data_ptr(), des.data_ptr(), off.data_ptr(), max_rows)
        ex.sync()
        off = off.cpu().numpy()
        want_off = np.concatenate([[0], np.cumsum(totals[:nframes])])
        assert off[nframes + 1] == want_off[-1]
        assert np.array_equal(off[:nframes + 1], np.minimum(want_off, max_rows))
        n = int(off[nframes])
        want_kp = np.concatenate([f.keypoints for f in per_frame[:nframes]])[:n]
        want_ori = np.concatenate([f.orientations for f in per_frame[:nframes]])[:n]
        want_des = np.concatenate([f.descriptors for f in per_frame[:nframes]])[:n]
        assert np.array_equal(kp[:n].cpu().numpy().view(np.uint32), want_kp.view(np.uint32))
        assert np.array_equal(ori[:n].cpu().numpy().view(np.uint32), want_ori.view(np.uint32))
        assert np.array_equal(des[:n].cpu().numpy().view(np.uint32), want_des.view(np.uint32))
        # nothing is written behind the packed rows
        assert bool((kp[n:] == -3.0).all()) and bool((des[n:] == -3.0).all())
    # detector only: no orientation / descriptor arrays
    ex.enqueue(px.data_ptr(), B, want_descriptors=False)
    kp = torch.zeros((B * ex.capacity, 4), device=dev)
    off = torch.zeros((B + 2,), dtype=torch.int32, device=dev)
    ex.pack_results(B, kp.data_ptr(), None, None, off.data_ptr(), B * ex.capacity)
    ex.sync()
    n = int(off[B].item())
    want = np.concatenate([ex.copy_results(f, want_descriptors=False)[1].keypoints for f in range(B)])
    assert n == len(want) and np.array_equal(kp[:n].cpu().numpy().view(np.uint32), want.view(np.uint32))
    with pytest.raises(_lib.MifError):
        ex.pack_results(B + 1, kp.data_ptr(), None, None, off.data_ptr(), 10)
    with pytest.raises(_lib.MifError):
        ex.pack_results(B, kp.data_ptr(), None, des.data_ptr(), off.data_ptr(), 10)
