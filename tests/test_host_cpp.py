"""The C++14 host layer (object_detection_amd/cpp): CPU unit tests of the compiled host code, and on the
GPU the two demo programs against the Python path / the checker."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "object_detection_amd", "cpp", "build")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _need(binary):
    path = os.path.join(BUILD, binary)
    if not os.path.exists(path):
        pytest.fail("%s missing: run `make -C object_detection_amd/cpp` (or __graft_entry__.build())" % path)
    return path


def test_host_layer_unit_tests(tmp_path):
    out = subprocess.run([_need("test_host"), str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host tests passed" in out.stdout


def test_demo_usage_and_argument_errors():
    exe = _need("demo_coopcl_sift")
    out = subprocess.run([exe, "-h"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "Usage" in out.stdout
    out = subprocess.run([exe, "-m", "a.pgm", "b.pgm"], capture_output=True, text=True, timeout=60)   # no -f
    assert "Usage" in out.stderr
    out = subprocess.run([_need("demo_coopcl_Surf"), "-i", "x.pgm", "-f", "7"], capture_output=True, text=True, timeout=60)
    assert "Usage" in out.stderr


@pytest.mark.gpu
def test_sift_match_demo_against_python_path(tmp_path):
    import object_detection_amd as oda
    import oracle_lib as orc
    rep = str(tmp_path / "sift.json")
    box, scene = os.path.join(GOLDEN, "box.pgm"), os.path.join(GOLDEN, "scene.pgm")
    out = subprocess.run([_need("demo_coopcl_sift"), "-m", box, scene, "-f", "0", "-o", rep],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    a, b = orc.read_pnm(box), orc.read_pnm(scene)
    _, fa = oda.SiftExtractor(a.shape[1], a.shape[0]).extract(a)
    _, fb = oda.SiftExtractor(b.shape[1], b.shape[0]).extract(b)
    assert "feats_object:\t%d" % len(fa.keypoints) in out.stdout
    assert "feats_scene:\t%d" % len(fb.keypoints) in out.stdout
    m = oda.Matcher(len(fa.descriptors), len(fb.descriptors))
    matches, dist, mn, mx = oda.match_sift_features(m, fa.descriptors, fb.descriptors)
    assert "Good matches: %d" % len(matches) in out.stdout
    r = json.load(open(rep))
    assert [g[:2] for g in r["good_matches"]] == matches.tolist()
    assert r["has_homography"] and r["inliers"] >= 8
    # the box lies inside the scene: its projected outline is a sane quadrilateral within the image margins
    c = np.array(r["scene_corners"])
    assert np.all(c[:, 0] > -100) and np.all(c[:, 0] < 612) and np.all(c[:, 1] > -100) and np.all(c[:, 1] < 484)
    area = 0.5 * abs(np.dot(c[:, 0], np.roll(c[:, 1], 1)) - np.dot(c[:, 1], np.roll(c[:, 0], 1)))
    assert 5000 < area < 120000


@pytest.mark.gpu
def test_surf_match_demo_against_checker(tmp_path):
    import oracle_lib as orc
    rep = str(tmp_path / "surf.json")
    box, scene = os.path.join(GOLDEN, "box.pgm"), os.path.join(GOLDEN, "scene.pgm")
    out = subprocess.run([_need("demo_coopcl_Surf"), "-m", box, scene, "-f", "1", "-o", rep],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    a, b = orc.read_pnm(box), orc.read_pnm(scene)
    Sa, Sb = orc.SurfOracle(a.shape[1], a.shape[0]), orc.SurfOracle(b.shape[1], b.shape[0])
    Sa.run(a), Sb.run(b)
    assert "feats_object:\t%d" % Sa.count() in out.stdout and "feats_scene:\t%d" % Sb.count() in out.stdout
    idx, dist = orc.match_top2(Sa.points()[:, 6:], Sb.points()[:, 6:])
    keep = dist[:, 0] < np.float32(0.75) * dist[:, 1]
    assert "Found stable matches:\t%d" % int(keep.sum()) in out.stdout
    r = json.load(open(rep))
    assert [g[:2] for g in r["good_matches"]] == np.stack([np.nonzero(keep)[0], idx[keep, 0]], 1).tolist()
    # the homography the reference's SURF overload lacks (kpmatch.h:270-320 stops at the matches): the box
    # lies inside the scene, so its projected outline is a sane quadrilateral within the image margins,
    # and it agrees with the outline the SIFT path finds for the same pair
    # (the ratio test leaves 7 matches for this pair; all of them agree with one homography)
    assert r["has_homography"] and 6 <= r["inliers"] <= len(r["good_matches"])
    c = np.array(r["scene_corners"])
    assert np.all(c[:, 0] > -100) and np.all(c[:, 0] < 612) and np.all(c[:, 1] > -100) and np.all(c[:, 1] < 484)
    area = 0.5 * abs(np.dot(c[:, 0], np.roll(c[:, 1], 1)) - np.dot(c[:, 1], np.roll(c[:, 0], 1)))
    assert 5000 < area < 120000
    rep2 = str(tmp_path / "sift.json")
    out = subprocess.run([_need("demo_coopcl_sift"), "-m", box, scene, "-f", "1", "-o", rep2],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    c2 = np.array(json.load(open(rep2))["scene_corners"])
    assert np.abs(c - c2).max() < 25.0, (c, c2)


@pytest.mark.gpu
@pytest.mark.parametrize("exe", ["demo_coopcl_sift", "demo_coopcl_Surf"])
def test_match_demo_repeat_loop_reports_the_same_result_with_times(tmp_path, exe):
    """`-m a b -f x -r N` repeats the pair N times on warm pipelines and adds the per-stage times to the
    report; the matches and the outline are those of the single run."""
    box, scene = os.path.join(GOLDEN, "box.pgm"), os.path.join(GOLDEN, "scene.pgm")
    reps = []
    for extra in ([], ["-r", "5"]):
        rep = str(tmp_path / ("rep%d.json" % len(reps)))
        out = subprocess.run([_need(exe), "-m", box, scene, "-f", "1", "-o", rep] + extra,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        reps.append(json.load(open(rep)))
    single, timed = reps
    assert timed["good_matches"] == single["good_matches"]
    assert timed["scene_corners"] == single["scene_corners"]
    t = timed["timing_ms"]
    assert t["repeats"] == 5
    stages = [t[k] for k in ("extract_object", "extract_scene", "convert", "match", "homography")]
    assert all(v >= 0.0 for v in stages) and t["extract_scene"] > 0.0 and t["match"] > 0.0
    # the stages are what a repeat consists of; its fastest run cannot take longer than the mean
    assert 0.0 < t["total_min"] <= t["total"] * 1.0001
    assert sum(stages) <= t["total"] * 1.05
    assert t["total"] < 1000.0


@pytest.mark.gpu
def test_stream_demo_on_raw_frames(tmp_path):
    from synth import blob_image
    frames = [blob_image(320, 240, seed=5, channels=3, shift=(2 * t, t)) for t in range(4)]
    raw = tmp_path / "clip.raw"
    raw.write_bytes(b"".join(f.tobytes() for f in frames))
    out = subprocess.run([_need("demo_coopcl_sift"), "-v", "%s:320x240x3" % raw, "-f", "0.3"],
                         capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Average time_ms:" in out.stdout and "Frames:\t4" in out.stdout
    assert os.path.exists(tmp_path / "sift_stage_times.csv")
    # the reference's record files (SIFT/sift.cpp:915-957), one line per extracted frame
    for stage in ("base_image", "scale_space", "detector", "orientation_descriptor"):
        for o in range(3):
            for kind, head in (("off", "offload"), ("cpu", "cpu_time"), ("gpu", "gpu_time")):
                lines = (tmp_path / ("%s%d0_%s.dat" % (stage, o, kind))).read_text().splitlines()
                assert lines[0] == "iteration\t" + head and len(lines) == 1 + 4, (stage, o, kind, lines)
                assert [ln.split("\t")[0] for ln in lines[1:]] == ["1", "2", "3", "4"]
    gpu_ms = [float(ln.split("\t")[1]) for ln in (tmp_path / "scale_space00_gpu.dat").read_text().splitlines()[1:]]
    assert all(0.0 < v < 1000.0 for v in gpu_ms)
    out = subprocess.run([_need("demo_coopcl_Surf"), "-v", "%s:320x240x3" % raw, "-f", "0.3"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "Average elapsed_time:" in out.stdout


@pytest.mark.gpu
def test_stream_demo_batched_equals_frame_by_frame(tmp_path):
    """`-v ... -b N` (frames through the GPU-resident pipeline in batches) reports the feature counts and the
    total of good matches that the frame-by-frame loop reports."""
    from synth import blob_image
    frames = [blob_image(320, 240, seed=5, channels=3, shift=(2 * t, t)) for t in range(7)]
    raw = tmp_path / "clip.raw"
    raw.write_bytes(b"".join(f.tobytes() for f in frames))

    def run(extra):
        out = subprocess.run([_need("demo_coopcl_sift"), "-v", "%s:320x240x3" % raw, "-f", "1"] + extra,
                             capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
        assert out.returncode == 0, out.stdout + out.stderr
        feats = [ln for ln in out.stdout.splitlines() if ln.startswith("Frame id<")]
        total = [ln for ln in out.stdout.splitlines() if ln.startswith("Frames:")][0].split("\t")
        return feats, total[1], total[3]

    one, n1, g1 = run([])
    for batch in ("3", "16"):
        many, nb, gb = run(["-b", batch])
        assert many == one and len(many) == 7
        assert (nb, gb) == (n1, g1) and int(g1) > 0
    # the sequence cut into blocks, one pipeline and host thread per block (`-g G`; on a one-GPU box the
    # blocks share GPU 0): every block but the first is extended by its predecessor frame, so the
    # records and the total of good matches are those of the single run
    for extra in (["-b", "3", "-g", "2"], ["-b", "2", "-g", "3"], ["-g", "7"], ["-b", "4", "-g", "50"]):
        many, nb, gb = run(extra)
        assert many == one and (nb, gb) == (n1, g1), extra
