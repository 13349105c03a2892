"""CPU tests of the checker itself: shared math, filter taps, and the restated
pipeline against (a) an independent numpy float64 restatement of the linear
stages and (b) the committed golden tables."""
import math
import os

import numpy as np
import pytest

import oracle_lib as orc
from synth import blob_image, sift_like_descriptors


def _ulp_err(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    return np.abs(got - want) / ulp


def test_shared_math_accuracy():
    L = orc.lib()
    rng = np.random.RandomState(1)
    xs = np.concatenate([rng.uniform(-80, 0, 4000), rng.uniform(0, 10, 1000)]).astype(np.float32)
    err = _ulp_err([L.orc_math_expf(float(x)) for x in xs], np.exp(xs.astype(np.float64)))
    assert err.max() <= 2.0
    xs = rng.uniform(-3, 3, 4000).astype(np.float32)
    err = _ulp_err([L.orc_math_exp2f(float(x)) for x in xs], np.exp2(xs.astype(np.float64)))
    assert err.max() <= 2.0
    a = rng.uniform(-1, 1, (6000, 2)).astype(np.float32)
    got = np.array([L.orc_math_atan2f(float(y), float(x)) for y, x in a])
    want = np.arctan2(a[:, 0].astype(np.float64), a[:, 1].astype(np.float64))
    assert np.abs(got - want).max() <= 5e-7
    assert L.orc_math_atan2f(0.0, 0.0) == 0.0
    assert abs(L.orc_math_atan2f(0.0, -1.0) - math.pi) < 1e-6
    xs = rng.uniform(-7, 7, 6000).astype(np.float32)
    s = np.array([L.orc_math_sinf(float(x)) for x in xs])
    c = np.array([L.orc_math_cosf(float(x)) for x in xs])
    assert np.abs(s - np.sin(xs.astype(np.float64))).max() <= 3e-7
    assert np.abs(c - np.cos(xs.astype(np.float64))).max() <= 3e-7


def math_directed_inputs():
    """Inputs where an elementary function is most likely to go wrong: axes and diagonals of atan2, both
    signs of zero, the switch points of the argument reductions, denormals, and the ends of the ranges
    the kernels use. Returns {op: (a, b or None)}."""
    f32 = np.float32
    tiny = [f32(1e-45), f32(-1e-45), f32(1.1754942e-38), f32(-1.1754942e-38), f32(1.17549435e-38), f32(1e-30)]
    mags = [f32(v) for v in (1e-45, 1.1754942e-38, 1e-30, 1e-8, 0.25, 1.0, 3.0, 255.0, 1e8, 3e38)]
    ys, xs = [], []
    for m in mags:                                   # axes, diagonals, both signs of zero
        for sy, sx in ((0.0, 1), (0.0, -1), (-0.0, 1), (-0.0, -1), (1, 0.0), (-1, 0.0), (1, -0.0), (-1, -0.0),
                       (1, 1), (1, -1), (-1, 1), (-1, -1)):
            ys.append(f32(sy) * m if sy not in (0.0, -0.0) else f32(sy))
            xs.append(f32(sx) * m if sx not in (0.0, -0.0) else f32(sx))
    t8 = f32(0.4142135679721832275390625)            # tan(pi/8): where atan2 changes its reduction
    for m in (f32(1.0), f32(7.0), f32(1e-20), f32(1e20)):
        for k in range(-4, 5):
            r = np.nextafter(t8, f32(2.0 if k > 0 else 0.0)) if k else t8
            for _ in range(abs(k) - 1):
                r = np.nextafter(r, f32(2.0 if k > 0 else 0.0))
            for sy, sx in ((1, 1), (-1, 1), (1, -1), (-1, -1)):
                ys += [f32(sy) * r * m, f32(sy) * m]
                xs += [f32(sx) * m, f32(sx) * r * m]
    ys += [f32(0.0), f32(-0.0), f32(0.0), f32(-0.0)]
    xs += [f32(0.0), f32(0.0), f32(-0.0), f32(-0.0)]
    e_in = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-38, -1e-10, 1e-10, -87.0, -86.999, -87.3365, -88.0, -103.0, -1e30,
                     88.0, 87.9999, 88.7, 0.34657359, -0.34657359, 0.35, -0.35, 0.6931472, -0.6931472, 1.0, -1.0]
                    + [0.5 * k * 0.6931471805599453 for k in range(-250, 255)]
                    + [float(np.nextafter(f32(0.5 * k * 0.6931471805599453), f32(d))) for k in range(-40, 41, 3)
                       for d in (-1e9, 1e9)], np.float32)
    e2_in = np.array([0.0, -0.0, 1e-45, 0.5, -0.5, 0.49999997, -0.49999997, 1.5, -1.5, 125.9, -125.9, 1 / 3.0, 2 / 3.0,
                      1.0, 4 / 3.0] + [k / 3.0 for k in range(-60, 61)] + [k + 0.5 for k in range(-120, 120, 7)],
                     np.float32)
    pio4 = 0.7853981633974483
    sc = [0.0, -0.0, 1e-45, -1e-45, 1e-38, 1e-20, -1e-20, 64.0, -64.0, 63.999996, -63.999996]
    for k in range(-82, 83):                         # every multiple of pi/4 up to |x| = 64 and its neighbours
        v = f32(k * pio4)
        sc += [float(v), float(np.nextafter(v, f32(1e9))), float(np.nextafter(v, f32(-1e9)))]
    sc = np.array([v for v in sc if abs(v) <= 64.0], np.float32)
    return {0: (e_in, None), 1: (e2_in, None), 2: (sc, None), 3: (sc, None),
            4: (np.array(ys, np.float32), np.array(xs, np.float32)),
            5: (e_in[(e_in >= -87.0) & (e_in <= 88.0)], None)}


def test_shared_math_directed_cases():
    """The elementary functions oracle and kernels share, on the inputs of math_directed_inputs, against
    libm in double: they are the one piece of arithmetic where the two sides cannot disagree, so their
    accuracy is pinned here and their GPU build is compared bit for bit in tests/test_sift_gpu.py."""
    cases = math_directed_inputs()
    with np.errstate(over="ignore", under="ignore", invalid="ignore"):
        a, _ = cases[0]
        got, want = orc.math_array(0, a), np.exp(a.astype(np.float64))
        fin = (a >= -87.0) & (a <= 88.0)
        assert _ulp_err(got[fin], want[fin]).max() <= 2.0
        assert np.all(got[a < -87.0] == 0.0) and np.all(np.isinf(got[a > 88.0]))       # documented range ends
        assert np.array_equal(orc.math_array(5, a[fin]).view(np.uint32), got[fin].view(np.uint32))
        a, _ = cases[1]
        assert _ulp_err(orc.math_array(1, a), np.exp2(a.astype(np.float64))).max() <= 2.0
        a, _ = cases[2]
        s, c = orc.math_array(2, a), orc.math_array(3, a)
        assert np.abs(s - np.sin(a.astype(np.float64))).max() <= 3e-7
        assert np.abs(c - np.cos(a.astype(np.float64))).max() <= 3e-7
        small = np.abs(a) < 1e-3                     # relative accuracy where sin x ~ x, denormals included
        assert np.array_equal(s[small], a[small]) and np.all(c[small] == 1.0)
        nz = small & (a != 0)                       # sin(-0) comes back as +0 (equal in value)
        assert np.all(np.signbit(s[nz]) == np.signbit(a[nz]))
        y, x = cases[4]
        got = orc.math_array(4, y, x).astype(np.float64)
        want = np.arctan2(y.astype(np.float64), x.astype(np.float64))
        both_zero = (y == 0) & (x == 0)
        assert np.all(got[both_zero] == 0.0)         # stated convention: atan2(+-0, +-0) = 0
        # y = -0 with x < 0: libm answers -pi, this implementation +pi (the same direction)
        diff = np.abs(got - want)[~both_zero]
        diff = np.minimum(diff, np.abs(diff - 2 * math.pi))
        assert diff.max() <= 5e-7
        assert np.all(np.abs(got) <= np.float64(np.float32(math.pi)))
        on_axis = (y == 0) & (x > 0)
        assert np.all(got[on_axis] == 0.0)
        up = (x == 0) & (y > 0)
        assert np.all(got[up] == np.float64(np.float32(math.pi / 2)))


def test_taps_match_reference_radii():
    S = orc.SiftOracle(64, 48)
    radii = []
    for l in range(6):
        t, fr = S.taps(l)
        radii.append(fr)
        assert len(t) == 2 * fr + 1
        assert abs(float(t.sum()) - 1.0) < 1e-6
        assert np.array_equal(t, t[::-1])
    # widths 13, 11, 13, 15, 19, 23 for sigma0 = 1.6, k = 2^(1/3) (sift.cpp:819-889)
    assert radii == [6, 5, 6, 7, 9, 11]


def _np_blur(img, taps, fr):
    """float64 restatement of kBlurH + kBlurV including the dropped last tap."""
    h, w = img.shape
    pad = np.zeros((h, w + 2 * fr), np.float64)
    pad[:, fr:fr + w] = img
    tmp = np.zeros((h, w), np.float64)
    for i in range(-fr, fr):
        tmp += float(taps[fr + i]) * pad[:, fr + i:fr + i + w]
    tmp = tmp.astype(np.float32).astype(np.float64)
    pad = np.zeros((h + 2 * fr, w), np.float64)
    pad[fr:fr + h] = tmp
    out = np.zeros((h, w), np.float64)
    for i in range(-fr, fr):
        out += float(taps[fr + i]) * pad[fr + i:fr + i + h]
    return out


@pytest.mark.parametrize("size", [(64, 48), (97, 61)])
def test_scale_space_against_numpy(size):
    w, h = size
    img = blob_image(w, h, seed=3)
    S = orc.SiftOracle(w, h)
    assert S.run(img, want_descriptors=False) == 0
    p = S.pyramid(0)
    cur = img.astype(np.float32) / np.float32(255.0)
    for l in range(6):
        t, fr = S.taps(l)
        want = _np_blur(cur.astype(np.float64), t, fr)
        assert np.abs(p[l] - want).max() < 2e-6
        cur = p[l]
    d = S.dog(0)
    for i in range(5):
        assert np.array_equal(d[i], p[i + 1] - p[i])
    # octave 1 base = level 3 of octave 0 sampled with the reference's indexing (sift.cpp:128-129)
    w1, h1 = S.dims(1)
    flat = p.reshape(-1)
    yy, xx = np.mgrid[0:h1, 0:w1]
    base1 = flat[(yy * 2) * (2 * w1) + xx * 2 + 3 * (2 * w1) * (2 * h1)]
    t, fr = S.taps(0)
    want = _np_blur(base1.astype(np.float64), t, fr)
    assert np.abs(S.pyramid(1)[0] - want).max() < 2e-6


def test_octave_geometry_and_limits():
    S = orc.SiftOracle(324, 223)
    assert [S.dims(o) for o in range(3)] == [(324, 223), (162, 111), (81, 55)]
    assert [S.max_features(o) for o in range(3)] == [722, 179, 44]


def test_flat_image_returns_minus_one():
    S = orc.SiftOracle(64, 64)
    assert S.run(np.full((64, 64), 128, np.uint8)) == -1
    assert S.count(0) == 0


def test_list_limit():
    from synth import dense_image
    S = orc.SiftOracle(256, 192)
    assert S.run(dense_image(256, 192, 0)) == 0
    assert S.raw_count(0) > S.max_features(0)
    assert S.count(0) == S.max_features(0)
    assert len(S.descriptors(0)) == S.max_features(0)


def test_keypoint_invariants():
    img = blob_image(256, 192, seed=1)
    S = orc.SiftOracle(256, 192)
    assert S.run(img) == 0
    for o in range(3):
        kp = S.keypoints(o)
        w, h = S.dims(o)
        assert len(kp) <= S.max_features(o)
        if not len(kp):
            continue
        assert np.all(np.isin(kp[:, 3], [1, 2, 3]))
        sc = 2.0 ** o
        assert np.all(kp[:, 0] / sc > 4) and np.all(kp[:, 0] / sc < w - 4)
        assert np.all(kp[:, 1] / sc > 4) and np.all(kp[:, 1] / sc < h - 4)
        ori = S.orientations(o)
        assert np.all((ori == -1) | ((ori >= -math.pi - 1e-6) & (ori <= math.pi + 1e-6)))
        d = S.descriptors(o)
        assert np.all(d >= 0) and np.all(d <= 255)


def test_color_path_close_to_gray():
    g = blob_image(96, 80, seed=5)
    bgr = np.repeat(g[:, :, None], 3, axis=2)
    A = orc.SiftOracle(96, 80, 1)
    B = orc.SiftOracle(96, 80, 3)
    assert A.run(g, False) == 0 and B.run(bgr, False) == 0
    assert np.abs(A.pyramid(0) - B.pyramid(0)).max() < 1e-6


GOLD = os.path.join(orc.GOLDEN, "sift_golden.npz")


@pytest.mark.parametrize("name", ["box", "scene", "blob_256x192", "blob_bgr_200x150", "dense_256x192", "blob_bgr_97x61"])
def test_oracle_matches_golden_tables(name):
    from make_golden import load_case
    gold = np.load(GOLD)
    img = load_case(name)
    h, w = img.shape[:2]
    S = orc.SiftOracle(w, h, 1 if img.ndim == 2 else 3)
    rc = S.run(img)
    assert rc == int(gold[name + "_rc"])
    if rc != 0:
        return
    kp, octs, ori, des = S.all_features()
    assert np.array_equal(kp, gold[name + "_kp"])
    assert np.array_equal(octs, gold[name + "_oct"])
    assert np.array_equal(ori, gold[name + "_ori"])
    assert np.array_equal(des, gold[name + "_des"])
    for o in range(3):
        assert np.array_equal(S.dog(o).view(np.uint32).sum(dtype=np.uint64), gold["%s_dogsum%d" % (name, o)])


def test_match_oracle_against_numpy():
    q = sift_like_descriptors(37, 1)
    db = sift_like_descriptors(211, 2)
    db[17] = q[3]
    db[40] = db[90] = q[5]                 # exact tie -> lowest index first
    idx, dist = orc.match_top2(q, db)
    d = np.sqrt(((q[:, None, :].astype(np.float64) - db[None].astype(np.float64)) ** 2).sum(-1))
    order = np.argsort(d, axis=1, kind="stable")[:, :2]
    assert np.array_equal(idx, order.astype(np.int32))
    assert np.allclose(dist, np.take_along_axis(d, order, 1), rtol=1e-6, atol=1e-4)
    assert idx[3, 0] == 17 and dist[3, 0] == 0.0
    assert list(idx[5]) == [40, 90]
    # dim < ld: trailing columns ignored
    q2, db2 = q.copy(), db.copy()
    q2[:, 127] = 1e6
    idx2, _ = orc.match_top2(q2, db2, dim=127)
    idx3, _ = orc.match_top2(q[:, :127].copy(), db[:, :127].copy())
    assert np.array_equal(idx2, idx3)
    # degenerate databases
    idx, dist = orc.match_top2(q, db[:1])
    assert np.all(idx[:, 0] == 0) and np.all(idx[:, 1] == -1) and np.all(np.isinf(dist[:, 1]))
    idx, dist = orc.match_top2(q, db[:0])
    assert np.all(idx == -1) and np.all(np.isinf(dist))


@pytest.mark.parametrize("channels", [1, 3])
def test_up_scale_equals_pixel_replication(channels):
    """up_scale resamples with ratio 2 by nearest neighbour (sift.cpp:134-210): every output pixel
    (x, y) takes input pixel (x >> 1, y >> 1), so the run must equal a plain run on the image with
    each pixel repeated 2 x 2."""
    img = blob_image(128, 96, seed=21, channels=channels)
    up = orc.SiftOracle(128, 96, channels, up_scale=True)
    rep = orc.SiftOracle(256, 192, channels)
    big = np.repeat(np.repeat(img, 2, axis=0), 2, axis=1)
    assert up.run(img) == rep.run(big)
    for o in range(3):
        assert up.dims(o) == rep.dims(o) and up.count(o) == rep.count(o)
        assert np.array_equal(up.pyramid(o).view(np.uint32), rep.pyramid(o).view(np.uint32))
    fu, fr = up.all_features(), rep.all_features()
    for a, b in zip(fu, fr):
        assert np.array_equal(a, b)
    assert up.count(0) > 0


def test_up_scale_odd_size_geometry():
    up = orc.SiftOracle(65, 47, 1, up_scale=True)
    assert [up.dims(o) for o in range(3)] == [(130, 94), (65, 47), (32, 23)]


def test_orientation_and_descriptor_against_second_restatement():
    """A few keypoints of every octave: the checker's orientation and descriptor must equal, bit for bit,
    the separate Python restatement of kOrientation / kDescriptor / kNormalize (tests/ref_restatement.py)."""
    import ref_restatement as ref
    img = blob_image(128, 96, seed=5)
    S = orc.SiftOracle(128, 96)
    assert S.run(img) == 0
    checked = 0
    for o in range(3):
        w, h = S.dims(o)
        pyr = S.pyramid(o).reshape(-1).copy()
        kps, ori, des = S.keypoints(o), S.orientations(o), S.descriptors(o)
        for i in range(min(4, len(kps))):
            th = ref.orientation(pyr, w, h, kps[i], o)
            assert th.view(np.uint32) == ori[i].view(np.uint32), (o, i, float(th), float(ori[i]))
            d = ref.descriptor(pyr, w, h, kps[i], ori[i], o)
            assert np.array_equal(d.view(np.uint32), des[i].view(np.uint32)), (o, i)
            checked += 1
    assert checked >= 6


def test_detector_against_second_restatement():
    """Every octave of two small images (one of them overflowing the list limit): the checker's keypoint
    list must equal, bit for bit and in order, the separate Python restatement of kDetector
    (tests/ref_restatement.py)."""
    import ref_restatement as ref
    from synth import dense_image
    total = 0
    for img in (blob_image(200, 150, seed=9), dense_image(160, 120, seed=3)):
        hh, ww = img.shape
        S = orc.SiftOracle(ww, hh)
        S.run(img, want_descriptors=False)
        for o in range(3):
            w, h = S.dims(o)
            kp = ref.detector(S.dog(o), w, h, o, S.max_features(o))
            got = S.keypoints(o)
            assert len(kp) == len(got), (o, len(kp), len(got))
            assert np.array_equal(kp.view(np.uint32), got.view(np.uint32)), o
            total += len(kp)
    assert total > 200


@pytest.mark.parametrize("ratio", [(2.0, 2.0), (0.5, 0.5), (1.7, 0.6), (3.0, 1.0)])
def test_resize_probes_against_numpy(ratio):
    """The checker's test_kResize / test_ColorToGray_Resize (SIFT/sift.cpp:2097-2144) against a numpy
    statement of kResize_NN_* (sift.cpp:136-206): source pixel = (trunc((y + 0.4995) / ry), trunc((x + 0.4995) / rx)),
    0 where that lies outside the image."""
    rx, ry = ratio
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    h, w = img.shape[:2]
    wd, hd = orc.resized_size(w, h, rx, ry)
    f = np.float32
    fy = (np.arange(hd, dtype=np.float32) + f(0.4995)) / f(ry)
    fx = (np.arange(wd, dtype=np.float32) + f(0.4995)) / f(rx)
    sy, sx = fy.astype(np.int32), fx.astype(np.int32)
    inside = (fy < h)[:, None] & (fx < w)[None, :]
    src = img[np.minimum(sy, h - 1)[:, None], np.minimum(sx, w - 1)[None, :]]
    want = np.where(inside[..., None], src, 0).astype(np.uint8)
    assert np.array_equal(orc.resize_u8(img, rx, ry), want)
    assert np.array_equal(orc.resize_u8(img[..., 0].copy(), rx, ry), want[..., 0])
    g = img.astype(np.float32)
    gray = ((g[..., 0] * f(0.114) + g[..., 1] * f(0.587)) + g[..., 2] * f(0.299)) * f(0.0039215686)
    wantg = np.where(inside, gray[np.minimum(sy, h - 1)[:, None], np.minimum(sx, w - 1)[None, :]], f(0))
    assert np.array_equal(orc.gray_resize(img, rx, ry).view(np.uint32), wantg.astype(np.float32).view(np.uint32))
