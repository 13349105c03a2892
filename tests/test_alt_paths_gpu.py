"""The library's diagnostic switches select the unfused forms of three stages (gradient planes by a
kernel of their own, base images written by the gray / decimation kernels, matching without the bf16
screening passes) and the general window loop of the descriptor kernel for every keypoint. Every form
must return the same bits as the default path."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, name, env_extra):
    out = str(tmp_path / (name + ".npz"))
    env = dict(os.environ)
    env.update(env_extra)
    subprocess.run([sys.executable, os.path.join(HERE, "_alt_paths_worker.py"), out], check=True, env=env, timeout=300)
    return np.load(out)


def test_unfused_paths_equal_the_default(tmp_path):
    ref = _run(tmp_path, "default", {})
    assert int(ref["gray_rc0"]) == 0 and len(ref["gray_kp0"]) > 20 and len(ref["bgr_kp1"]) > 20
    alt = _run(tmp_path, "alt", {"MIF_NO_FUSED_GRAD": "1", "MIF_NO_FUSED_BASE": "1", "MIF_MATCH_NO_SCREEN": "1",
                                 "MIF_DESC_PLAIN": "1"})
    assert sorted(ref.files) == sorted(alt.files)
    for k in ref.files:
        a, b = ref[k], alt[k]
        assert a.shape == b.shape, k
        assert a.tobytes() == b.tobytes(), k


def test_descriptor_window_loops(tmp_path):
    """The descriptor kernel has three window loops (general; streamlined walk; mark the pixels that pass the
    cell test, then visit only those: the default): all must give the same bits. MIF_DESC_PLAIN=1 (general
    loop) is part of test_unfused_paths_equal_the_default."""
    ref = _run(tmp_path, "default", {})
    alt = _run(tmp_path, "walk", {"MIF_DESC_MODE": "1"})
    for k in ref.files:
        assert ref[k].shape == alt[k].shape and ref[k].tobytes() == alt[k].tobytes(), k


def test_staging_and_surf_switches(tmp_path):
    """MIF_NO_WIDE_STAGING=1 (a BGR frame fetched pixel by pixel instead of four pixels per load) and
    MIF_SURF_NO_STAGED_HESSIAN=1 (all Hessian layers straight from memory) with another grid for the SURF
    point kernels (MIF_SURF_BLOCKS): the default's bits."""
    ref = _run(tmp_path, "default", {})
    assert any(k.startswith("surf_") for k in ref.files)
    alt = _run(tmp_path, "plain", {"MIF_NO_WIDE_STAGING": "1", "MIF_SURF_NO_STAGED_HESSIAN": "1",
                                   "MIF_SURF_BLOCKS": "72"})
    assert sorted(ref.files) == sorted(alt.files)
    for k in ref.files:
        assert ref[k].shape == alt[k].shape and ref[k].tobytes() == alt[k].tobytes(), k


@pytest.mark.parametrize("form", ["fused", "twopass"])
def test_surf_integral_image_forms(tmp_path, form):
    """MIF_SURF_INTEGRAL forces one form of the integral image at any size: both passes in one launch with
    the bands of a frame handing their column totals down (what batches use by themselves), or a launch
    per pass (what single frames use). Integral image, response layers and points: the default's bits."""
    ref = _run(tmp_path, "default", {})
    alt = _run(tmp_path, form, {"MIF_SURF_INTEGRAL": form})
    for k in ref.files:
        if k.startswith("surf_"):
            assert ref[k].shape == alt[k].shape and ref[k].tobytes() == alt[k].tobytes(), k


@pytest.mark.parametrize("mask", ["0", "7"])
def test_scale_space_levels_singly_or_in_pairs(tmp_path, mask):
    """MIF_BLUR_PAIRS selects which levels of an octave are built two per launch (default: levels 0 and 1
    only): one level per launch (0) and all three pairs (7) must give the default's bits."""
    ref = _run(tmp_path, "default", {})
    alt = _run(tmp_path, "pairs" + mask, {"MIF_BLUR_PAIRS": mask})
    for k in ref.files:
        assert ref[k].shape == alt[k].shape and ref[k].tobytes() == alt[k].tobytes(), k


@pytest.mark.parametrize("env", [{"MIF_MATCH_SCREEN_V1": "1"}, {"MIF_MATCH_SCREEN_WIDE": "1"},
                                 {"MIF_MATCH_SCREEN_WIDE": "0"}, {"MIF_MATCH_SCREEN_TURNS": "1"},
                                 {"MIF_MATCH_SCREEN_TURNS": "0"},
                                 {"MIF_MATCH_SCREEN_TURNS": "1", "MIF_MATCH_SAMPLE": "3"},
                                 {"MIF_MATCH_SCREEN_TURNS": "0", "MIF_MATCH_SAMPLE": "5"}])
def test_screening_kernel_forms(tmp_path, env):
    """The screening passes exist in four forms (32 queries per wave with the norms added in the epilogue;
    64 queries per wave with the norms inside the contraction, in blocks of 4 or of 8 waves; the same
    contraction over a bf16 image of the database rows copied straight into LDS, which large searches use
    by themselves and MIF_MATCH_SCREEN_TURNS=1 forces at any size): each must give the default's bits, also
    when the first pass looks at every n-th chunk of rows only (MIF_MATCH_SAMPLE: what large searches do,
    with a looser limit, longer candidate lists and queries handed to the overflow kernel)."""
    ref = _run(tmp_path, "default", {})
    alt = _run(tmp_path, "screen", env)
    for k in ("match_idx", "match_dist"):
        assert ref[k].shape == alt[k].shape and ref[k].tobytes() == alt[k].tobytes(), k
