"""CPU tests of the SURF checker against independent numpy restatements and the golden table."""
import os

import numpy as np
import pytest

import oracle_lib as orc
from synth import blob_image


def _kogge_stone_scan(row):
    """numpy float32 restatement of integralImage_kernels.cl:63-152 for one row."""
    out = np.zeros_like(row)
    prev = np.float32(0)
    for x0 in range(0, len(row), 64):
        blk = np.zeros(64, np.float32)
        seg = row[x0:x0 + 64]
        blk[:len(seg)] = seg
        for d in (1, 2, 4, 8, 16, 32):
            sh = np.zeros(64, np.float32)
            sh[d:] = blk[:-d]
            blk = (blk + sh).astype(np.float32)
        out[x0:x0 + 64] = (blk[:len(seg)] + prev).astype(np.float32)
        prev = np.float32(prev + blk[63])
    return out


@pytest.mark.parametrize("size", [(97, 61), (200, 150)])
def test_integral_image_against_numpy(size):
    w, h = size
    img = blob_image(w, h, seed=4)
    S = orc.SurfOracle(w, h)
    S.run(img)
    gray = (img.astype(np.float64) * (1.0 / 255.0)).astype(np.float32)
    rows = np.stack([_kogge_stone_scan(gray[y]) for y in range(h)])
    want = np.stack([_kogge_stone_scan(np.ascontiguousarray(rows[:, x])) for x in range(w)], axis=1)
    assert np.array_equal(S.integral(), want)
    exact = np.cumsum(np.cumsum(gray.astype(np.float64), axis=0), axis=1)
    assert np.abs(S.integral() - exact).max() / exact.max() < 1e-5


def test_layer_geometry_and_response_formula():
    img = orc.read_pnm(os.path.join(orc.GOLDEN, "box.pgm"))
    h, w = img.shape
    S = orc.SurfOracle(w, h)
    n = S.run(img)
    assert n == S.count() > 50
    assert [S.layer_info(i) for i in range(8)] == [
        (64, 44, 5, 9), (64, 44, 5, 15), (64, 44, 5, 21), (64, 44, 5, 27),
        (32, 22, 10, 39), (32, 22, 10, 51), (16, 11, 20, 75), (16, 11, 20, 99)]
    # one interior response value recomputed from the integral image in float64
    I = S.integral().astype(np.float64)

    def box(r, c, rows, cols):
        r1, c1 = min(r, h) - 1, min(c, w) - 1
        r2, c2 = min(r + rows, h) - 1, min(c + cols, w) - 1
        g = lambda rr, cc: I[rr, cc] if rr >= 0 and cc >= 0 else 0.0
        return max(0.0, g(r1, c1) - g(r1, c2) - g(r2, c1) + g(r2, c2))

    resp, lap = S.layer(1)
    idy, idx, step, f = 20, 30, 5, 15
    r, c, l, b = idy * step, idx * step, f // 3, (f - 1) // 2 + 1
    dxx = box(r - l + 1, c - b, 2 * l - 1, f) - 3 * box(r - l + 1, c - l // 2, 2 * l - 1, l)
    dyy = box(r - b, c - l + 1, f, 2 * l - 1) - 3 * box(r - l // 2, c - l + 1, l, 2 * l - 1)
    dxy = box(r - l, c + 1, l, l) + box(r + 1, c - l, l, l) - box(r - l, c - l, l, l) - box(r + 1, c + 1, l, l)
    dxx, dyy, dxy = dxx / f / f, dyy / f / f, dxy / f / f
    assert abs(resp[idy, idx] - (dxx * dyy - 0.81 * dxy * dxy)) < 1e-5
    assert lap[idy, idx] == (1 if dxx + dyy >= 0 else 0)


def test_point_invariants():
    img = orc.read_pnm(os.path.join(orc.GOLDEN, "scene.pgm"))
    h, w = img.shape
    S = orc.SurfOracle(w, h)
    S.run(img)
    P = S.points()
    assert len(P) == S.raw_count() <= 8000
    assert np.all((P[:, 0] >= 0) & (P[:, 0] < w) & (P[:, 1] >= 0) & (P[:, 1] < h))
    assert np.all(P[:, 2] > 1.0)
    assert np.all((P[:, 3] >= 0) & (P[:, 3] <= 2 * np.pi + 1e-5))
    assert np.all(np.isin(P[:, 4], [0, 1]))
    assert np.allclose(np.linalg.norm(P[:, 6:], axis=1), 1.0, atol=1e-5)


def test_flat_image_has_no_points():
    S = orc.SurfOracle(128, 96)
    assert S.run(np.full((96, 128), 100, np.uint8)) == 0


@pytest.mark.parametrize("name", ["box", "scene", "blob_bgr_200x150"])
def test_surf_oracle_matches_golden(name):
    from make_golden import load_case
    gold = np.load(os.path.join(orc.GOLDEN, "surf_golden.npz"))
    img = load_case(name)
    h, w = img.shape[:2]
    S = orc.SurfOracle(w, h)
    S.run(img)
    assert np.array_equal(S.points().view(np.uint32), gold[name].view(np.uint32))
    assert np.array_equal(S.integral().view(np.uint32).sum(dtype=np.uint64), gold[name + "_intsum"])


@pytest.mark.parametrize("name", ["box", "scene"])
def test_point_selection_against_second_restatement(name):
    """Positions, scales and laplacians of all interest points, in order, against the separate Python
    restatement of selectIpoints / non_max_supression (tests/ref_restatement.py), bit for bit."""
    import ref_restatement as ref
    img = orc.read_pnm(os.path.join(orc.GOLDEN, name + ".pgm"))
    h, w = img.shape
    S = orc.SurfOracle(w, h)
    S.run(img)
    layers = []
    for i in range(8):
        lw, lh, step, flt = S.layer_info(i)
        r, l = S.layer(i)
        layers.append((r, l, step, flt))
    want = ref.surf_select_points(layers)
    got = S.points()
    assert len(want) == len(got) == S.count() and len(want) > 50
    for (x, y, sc, lap), row in zip(want, got):
        assert np.float32(x).view(np.uint32) == row[0].view(np.uint32)
        assert np.float32(y).view(np.uint32) == row[1].view(np.uint32)
        assert np.float32(sc).view(np.uint32) == row[2].view(np.uint32)
        assert lap == int(row[4])


@pytest.mark.parametrize("name", ["box", "scene"])
def test_orientation_and_descriptor_against_second_restatement(name):
    """Orientation and the 64 descriptor values of a spread of interest points against the separate
    Python restatement of getOrientationStep1/2, createDescriptors and normalizeDescriptors
    (tests/ref_restatement.py), bit for bit. The restatement gets the oracle's integral image and point
    positions (both pinned by the tests above) and, for the descriptor, the oracle's orientation."""
    import ref_restatement as ref
    img = orc.read_pnm(os.path.join(orc.GOLDEN, name + ".pgm"))
    h, w = img.shape
    S = orc.SurfOracle(w, h)
    S.run(img)
    integral = S.integral()
    pts = S.points()
    assert len(pts) > 50
    picks = sorted(set(np.linspace(0, len(pts) - 1, 12).astype(int).tolist()))
    for i in picks:
        x, y, scale, ori = pts[i, 0], pts[i, 1], pts[i, 2], pts[i, 3]
        want_ori = ref.surf_orientation(integral, x, y, scale)
        assert np.float32(want_ori).view(np.uint32) == ori.view(np.uint32), (i, want_ori, ori)
        want_desc = ref.surf_descriptor(integral, x, y, scale, ori)
        assert np.array_equal(want_desc.view(np.uint32), pts[i, 6:70].view(np.uint32)), i
