"""How far the checker's results depend on the math library.

The reference leaves exp / atan2 / sin / cos to the OpenCL device it runs on (built-ins under
-cl-fast-relaxed-math, SIFT/sift.cpp:1369-1370; SURF kernels likewise), so its own results differ from
device to device by what those built-ins round differently. The parity tests compare the HIP kernels
with a checker that shares their implementations of these functions (csrc/mif_math.h); here the checker
is built a second time with the C library's double-precision functions rounded to float
(oracle/orc_math.h, -DORC_MATH_LIBM) and both builds run on the same images. What is asserted is what a
user of the reference can rely on whichever library computes the transcendental functions: the same
keypoints in the same order, positions / scales / orientations equal to a few float roundings,
descriptors equal to a few parts in a million of their length, and the same good matches.
"""
import os

import numpy as np
import pytest

import oracle_lib as orc
from synth import blob_image

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _images():
    yield "box", orc.read_pnm(os.path.join(GOLDEN, "box.pgm"))
    yield "scene", orc.read_pnm(os.path.join(GOLDEN, "scene.pgm"))
    yield "blobs", blob_image(640, 480, seed=2)
    yield "blobs_bgr", blob_image(321, 243, seed=7, channels=3)


def _sift(img, library):
    h, w = img.shape[:2]
    S = orc.SiftOracle(w, h, 1 if img.ndim == 2 else 3, library=library)
    S.run(img)
    counts = [S.count(o) for o in range(3)]
    return counts, S.all_features()


@pytest.mark.parametrize("name,img", list(_images()), ids=[n for n, _ in _images()])
def test_sift_features_do_not_depend_on_the_math_library(name, img):
    ca, (ka, oa, ra, da) = _sift(img, None)
    cb, (kb, ob, rb, db) = _sift(img, orc.lib_libm())
    assert ca == cb and sum(ca) > 50
    # position, scale: refined by a 3 x 3 solve of finite differences, no transcendental function but
    # the level's sigma (exp2)
    assert np.abs(ka - kb).max() <= 2e-6
    # orientation: a parabola through three bins of a histogram whose weights are exp() values
    assert np.abs(ra - rb).max() <= 5e-6
    # descriptors have length 512: entries agree to 1e-5 of that
    assert np.allclose(np.linalg.norm(da, axis=1), 512.0, atol=1e-2)
    assert np.abs(da - db).max() <= 5e-3


def test_sift_good_matches_do_not_depend_on_the_math_library():
    box = orc.read_pnm(os.path.join(GOLDEN, "box.pgm"))
    scene = orc.read_pnm(os.path.join(GOLDEN, "scene.pgm"))
    good = []
    for library in (None, orc.lib_libm()):
        _, (_, _, _, da) = _sift(box, library)
        _, (_, _, _, db) = _sift(scene, library)
        idx, dist = orc.match_top2(da, db)
        keep = dist[:, 0] < np.float32(0.75) * dist[:, 1]
        good.append(np.stack([np.nonzero(keep)[0], idx[keep, 0]], 1))
    assert len(good[0]) >= 8
    assert np.array_equal(good[0], good[1])


@pytest.mark.parametrize("name,img", list(_images()), ids=[n for n, _ in _images()])
def test_surf_points_do_not_depend_on_the_math_library(name, img):
    h, w = img.shape[:2]
    pts = []
    for library in (None, orc.lib_libm()):
        S = orc.SurfOracle(w, h, library=library)
        S.run(img)
        pts.append(S.points())
    assert pts[0].shape == pts[1].shape and len(pts[0]) > 20
    # x, y, scale, orientation, laplacian, ..., then 64 descriptor entries of a unit vector
    assert np.abs(pts[0] - pts[1]).max() <= 5e-6
