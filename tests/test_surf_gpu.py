"""GPU parity: the HIP SURF path (through the C ABI) against the CPU checker and the golden table.
Bar: counts and laplacians exact, every fp32 output bit-identical (documented fallback tolerance of
the reference's own comparison protocol, SURF/coopcl_surf/src/utils.cpp:127-185, is 0.1)."""
import os

import numpy as np
import pytest

import object_detection_amd as oda
import oracle_lib as orc
from make_golden import load_case
from synth import blob_image

pytestmark = pytest.mark.gpu


def _compare(img):
    h, w = img.shape[:2]
    c = 1 if img.ndim == 2 else 3
    S = orc.SurfOracle(w, h)
    n = S.run(img)
    ex = oda.SurfExtractor(w, h, c)
    pts, raw = ex.run(img)
    assert len(pts) == n and raw == S.raw_count()
    assert np.array_equal(ex.integral().view(np.uint32), S.integral().view(np.uint32))
    for i in range(8):
        assert ex.layer_info(i) == S.layer_info(i)
        r_g, l_g = ex.layer(i)
        r_o, l_o = S.layer(i)
        assert np.array_equal(r_g.view(np.uint32), r_o.view(np.uint32)), i
        assert np.array_equal(l_g, l_o), i
    P = S.points()
    if n:
        assert np.array_equal(pts["x"].view(np.uint32), P[:, 0].view(np.uint32))
        assert np.array_equal(pts["y"].view(np.uint32), P[:, 1].view(np.uint32))
        assert np.array_equal(pts["scale"].view(np.uint32), P[:, 2].view(np.uint32))
        assert np.array_equal(pts["laplacian"], P[:, 4].astype(np.int32))
        assert np.array_equal(pts["orientation"].view(np.uint32), P[:, 3].view(np.uint32))
        assert np.array_equal(pts["descriptor"].view(np.uint32), P[:, 6:].view(np.uint32))
    return pts, P


@pytest.mark.parametrize("name", ["box", "scene", "blob_bgr_200x150", "blob_256x192"])
def test_surf_parity(name):
    pts, P = _compare(load_case(name))
    gold = np.load(os.path.join(orc.GOLDEN, "surf_golden.npz"))
    if name in gold:
        assert np.array_equal(pts["descriptor"], gold[name][:, 6:])


@pytest.mark.parametrize("size", [(64, 64), (65, 47), (333, 129), (20, 20), (1, 1), (12, 300)])
def test_surf_ragged_sizes(size):
    w, h = size
    _compare(blob_image(w, h, seed=w + 7 * h))


def test_surf_flat_image_and_reuse():
    ex = oda.SurfExtractor(256, 192)
    pts, raw = ex.run(np.full((192, 256), 50, np.uint8))
    assert len(pts) == 0 and raw == 0
    img = blob_image(256, 192, seed=1)
    S = orc.SurfOracle(256, 192)
    S.run(img)
    pts, _ = ex.run(img)
    assert np.array_equal(pts["descriptor"].view(np.uint32), S.points()[:, 6:].view(np.uint32))


def test_surf_module_interface_and_matching():
    box, scene = load_case("box"), load_case("scene")
    assert len(oda.surf.run(box)) == 0 or oda.surf._initialized      # empty until the driver is initialised
    assert oda.surf.initialize_surf_driver([("ignored.cl", "scan")], offload=0.5)
    a, b = oda.surf.run(box), oda.surf.run(scene)
    Sa, Sb = orc.SurfOracle(324, 223), orc.SurfOracle(512, 384)
    Sa.run(box), Sb.run(scene)
    assert np.array_equal(a["descriptor"], Sa.points()[:, 6:]) and np.array_equal(b["descriptor"], Sb.points()[:, 6:])
    m = oda.Matcher(len(a), len(b))
    good, dist = oda.match_ratio_test(m, a["descriptor"], b["descriptor"])
    idx_o, dist_o = orc.match_top2(Sa.points()[:, 6:], Sb.points()[:, 6:])
    keep = dist_o[:, 0] < np.float32(0.75) * dist_o[:, 1]
    assert np.array_equal(good[:, 0], np.nonzero(keep)[0]) and np.array_equal(good[:, 1], idx_o[keep, 0])
    assert len(good) >= 5


def test_surf_full_hd():
    img = blob_image(1920, 1080, seed=1234, channels=3)
    pts, P = _compare(img)
    assert len(pts) > 1000


@pytest.mark.parametrize("channels", [1, 3])
def test_surf_batches_equal_single_frames(channels):
    """Five different frames through one batched handle (a full batch of 4, then a partial one of 1 + a
    repeat): integral image, counts and every Ipoint field of every frame equal the checker's."""
    import torch
    w, h = 333, 129
    frames = [blob_image(w, h, seed=40 + i, channels=channels, shift=(3 * i, i)) for i in range(5)]
    frames[2] = np.full_like(frames[2], 77)                 # a flat frame: no interest points
    ex = oda.SurfExtractor(w, h, channels, batch=4)
    dev = torch.device("cuda:0")
    S = orc.SurfOracle(w, h)
    with pytest.raises(Exception):
        ex.stage_times()                                    # nothing has been timed yet
    ex.set_timing(True)
    for lo, hi in ((0, 4), (4, 5), (1, 3)):
        px = torch.from_numpy(np.stack(frames[lo:hi])).to(dev)
        torch.cuda.synchronize()
        ex.enqueue_batch(px.data_ptr(), hi - lo)
        t = ex.stage_times()                                # per stage, HIP events on the handle's stream
        assert t.shape == (len(ex.STAGES),) and np.all(t >= 0) and 0 < t.sum() < 1000
        for f in range(hi - lo):
            n = S.run(frames[lo + f])
            pts, raw = ex.copy_points(f)
            assert len(pts) == n and raw == S.raw_count(), (lo, f)
            assert np.array_equal(ex.integral(f).view(np.uint32), S.integral().view(np.uint32)), (lo, f)
            if n:
                P = S.points()
                got = np.concatenate([pts["x"][:, None], pts["y"][:, None], pts["scale"][:, None],
                                      pts["orientation"][:, None]], axis=1)
                assert np.array_equal(got.view(np.uint32), P[:, :4].view(np.uint32)), (lo, f)
                assert np.array_equal(pts["laplacian"], P[:, 4].astype(np.int32))
                assert np.array_equal(pts["descriptor"].view(np.uint32), P[:, 6:].view(np.uint32)), (lo, f)
    lib = oda.load_library()
    import ctypes as C
    from object_detection_amd import _lib
    assert lib.mif_surf_enqueue_batch(ex._h, C.c_void_p(px.data_ptr()), 5) == _lib.MIF_ERR_CAPACITY
    assert lib.mif_surf_batch(ex._h) == 4


@pytest.mark.parametrize("w,h,channels", [(200, 130, 3), (203, 129, 3), (64, 193, 1), (131, 140, 1)])
def test_surf_batch_with_both_passes_of_the_integral_image_in_one_launch(w, h, channels):
    """Batches of 192 bands of 64 rows and more take the integral image's one-launch form (bands of a frame
    hand their column totals down through memory): 64 and 70 frames of 3 / 4 bands, widths that are and are
    not multiples of four (four pixels per load, or one), then a smaller batch on the same handle (below
    the limit: a launch per pass) and the large one again. Integral image and points of probed frames equal
    the checker's."""
    import torch
    B = 70
    frames = np.stack([blob_image(w, h, seed=300 + i, channels=channels, shift=(i % 7, i % 5)) for i in range(B)])
    frames[5] = 0
    frames[6] = 255
    dev = torch.device("cuda:0")
    px = torch.from_numpy(frames).to(dev)
    ex = oda.SurfExtractor(w, h, channels, batch=B)
    S = orc.SurfOracle(w, h)
    for nf in (64, 70, 9, 70):
        assert ((h + 63) // 64) * nf >= 192 or nf == 9
        ex.enqueue_batch(px.data_ptr(), nf)
        for f in sorted({0, 5, 6, nf // 2, nf - 1}):
            n = S.run(frames[f])
            assert np.array_equal(ex.integral(f).view(np.uint32), S.integral().view(np.uint32)), (nf, f)
            pts, raw = ex.copy_points(f)
            assert len(pts) == n and raw == S.raw_count(), (nf, f)
            if n:
                assert np.array_equal(pts["descriptor"].view(np.uint32), S.points()[:, 6:].view(np.uint32)), (nf, f)
    ex.close()


def test_surf_batch_whose_bands_give_up_waiting_is_built_again(monkeypatch):
    """MIF_SURF_INTEGRAL_TEST_FAULT=1 makes the bands of the one-launch integral image wait for a launch
    number nobody writes: every band gives up (a bounded wait), and the launch-per-pass kernels enqueued
    behind build the batch again. Results equal the checker's, the handle counts the event, and the next
    batch (switch off) goes through the one-launch form untroubled."""
    import torch
    w, h, B = 200, 130, 64
    frames = np.stack([blob_image(w, h, seed=500 + i, channels=3) for i in range(B)])
    px = torch.from_numpy(frames).to(torch.device("cuda:0"))
    ex = oda.SurfExtractor(w, h, 3, batch=B)
    S = orc.SurfOracle(w, h)
    assert ex.integral_fallbacks() == 0
    for forced, expect in ((True, 1), (False, 1)):
        if forced:
            monkeypatch.setenv("MIF_SURF_INTEGRAL_TEST_FAULT", "1")
        else:
            monkeypatch.delenv("MIF_SURF_INTEGRAL_TEST_FAULT")
        ex.enqueue_batch(px.data_ptr(), B)
        for f in (0, 31, 63):
            n = S.run(frames[f])
            assert np.array_equal(ex.integral(f).view(np.uint32), S.integral().view(np.uint32)), (forced, f)
            pts, raw = ex.copy_points(f)
            assert len(pts) == n and raw == S.raw_count(), (forced, f)
        assert ex.integral_fallbacks() == expect
    ex.close()


def test_surf_tall_image_column_scan():
    """More rows than the tiled column scan's table of block totals holds (8192): the one-lane-per-column
    form takes over; the integral image still equals the checker's."""
    w, h = 70, 8300
    img = blob_image(w, h, seed=5)
    S = orc.SurfOracle(w, h)
    n = S.run(img)
    ex = oda.SurfExtractor(w, h, 1)
    pts, raw = ex.run(img)
    assert np.array_equal(ex.integral().view(np.uint32), S.integral().view(np.uint32))
    assert len(pts) == n
