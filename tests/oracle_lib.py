"""ctypes access to the CPU checker library built from oracle/ (test infrastructure only)."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "build", "liboracle.so")
# the same checker with the C library's exp / atan2 / sin / cos (oracle/orc_math.h)
ORACLE_LIBM_SO = os.path.join(ROOT, "oracle", "build", "liboracle_libm.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

_libs = {}


def lib(path=ORACLE_SO):
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError("%s missing: run `make -C oracle` (or __graft_entry__.build())" % path)
        L = C.CDLL(path)
        L.orc_sift_create.restype = C.c_void_p
        L.orc_sift_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_sift_create_up.restype = C.c_void_p
        L.orc_sift_create_up.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_sift_destroy.argtypes = [C.c_void_p]
        L.orc_sift_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        for name in ("orc_sift_octave_width", "orc_sift_octave_height", "orc_sift_max_features",
                     "orc_sift_count", "orc_sift_raw_count"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
            getattr(L, name).restype = C.c_int
        for name in ("orc_sift_pyramid", "orc_sift_dog", "orc_sift_keypoints", "orc_sift_orientations",
                     "orc_sift_descriptors"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
            getattr(L, name).restype = C.POINTER(C.c_float)
        L.orc_sift_taps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int]
        L.orc_sift_taps.restype = C.c_int
        for name in ("orc_math_expf", "orc_math_exp2f", "orc_math_sinf", "orc_math_cosf"):
            getattr(L, name).argtypes = [C.c_float]
            getattr(L, name).restype = C.c_float
        L.orc_math_atan2f.argtypes = [C.c_float, C.c_float]
        L.orc_math_atan2f.restype = C.c_float
        L.orc_math_array.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_math_array.restype = None
        L.orc_match_top2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_match_top2.restype = C.c_int
        L.orc_match_top2_scores.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
        L.orc_match_top2_scores.restype = C.c_int
        L.orc_resize_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.orc_resize_u8.restype = C.c_int
        L.orc_gray_resize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.orc_gray_resize.restype = C.c_int
        L.orc_surf_create.restype = C.c_void_p
        L.orc_surf_create.argtypes = [C.c_int, C.c_int]
        L.orc_surf_destroy.argtypes = [C.c_void_p]
        L.orc_surf_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_surf_count.argtypes = [C.c_void_p]
        L.orc_surf_raw_count.argtypes = [C.c_void_p]
        L.orc_surf_integral.argtypes = [C.c_void_p]
        L.orc_surf_integral.restype = C.POINTER(C.c_float)
        L.orc_surf_layer_info.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 4
        L.orc_surf_layer_responses.argtypes = [C.c_void_p, C.c_int]
        L.orc_surf_layer_responses.restype = C.POINTER(C.c_float)
        L.orc_surf_layer_laplacians.argtypes = [C.c_void_p, C.c_int]
        L.orc_surf_layer_laplacians.restype = C.POINTER(C.c_int)
        L.orc_surf_points.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_get_threads.restype = C.c_int
        _libs[path] = L
    return _libs[path]


def lib_libm():
    return lib(ORACLE_LIBM_SO)


def math_array(op, a, b=None):
    """mif_math.h as gcc compiles it, over arrays; op: 0 exp, 1 exp2, 2 sin, 3 cos, 4 atan2(a, b),
    5 exp without range guards."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(a if b is None else b, np.float32)
    out = np.empty_like(a)
    lib().orc_math_array(op, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size)
    return out


def read_pnm(path):
    """Binary PGM (P5) or PPM (P6) -> uint8 array [h, w] or [h, w, 3] (RGB order as stored)."""
    with open(path, "rb") as f:
        data = f.read()
    toks, pos = [], 0
    while len(toks) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            while data[pos:pos + 1] != b"\n":
                pos += 1
            continue
        start = pos
        while not data[pos:pos + 1].isspace():
            pos += 1
        toks.append(data[start:pos])
    pos += 1
    magic, w, h, maxv = toks[0], int(toks[1]), int(toks[2]), int(toks[3])
    assert maxv == 255
    ch = {b"P5": 1, b"P6": 3}[magic]
    arr = np.frombuffer(data, dtype=np.uint8, count=w * h * ch, offset=pos)
    return arr.reshape(h, w) if ch == 1 else arr.reshape(h, w, 3)


def _arr(ptr, n, dtype=np.float32):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


class SiftOracle:
    """One CPU run of the restated reference pipeline, with every intermediate image exposed."""

    def __init__(self, width, height, channels=1, up_scale=False, library=None):
        self.L = library or lib()
        self.h = self.L.orc_sift_create_up(width, height, channels, 1 if up_scale else 0)
        if not self.h:
            raise ValueError("bad image geometry")
        self.width, self.height, self.channels = width, height, channels

    def close(self):
        if self.h:
            self.L.orc_sift_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def run(self, pixels, want_descriptors=True):
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        assert px.size == self.width * self.height * self.channels
        return self.L.orc_sift_run(self.h, px.ctypes.data, 1 if want_descriptors else 0)

    def dims(self, o):
        return self.L.orc_sift_octave_width(self.h, o), self.L.orc_sift_octave_height(self.h, o)

    def max_features(self, o):
        return self.L.orc_sift_max_features(self.h, o)

    def count(self, o):
        return self.L.orc_sift_count(self.h, o)

    def raw_count(self, o):
        return self.L.orc_sift_raw_count(self.h, o)

    def pyramid(self, o):
        w, h = self.dims(o)
        return _arr(self.L.orc_sift_pyramid(self.h, o), 6 * w * h).reshape(6, h, w)

    def dog(self, o):
        w, h = self.dims(o)
        return _arr(self.L.orc_sift_dog(self.h, o), 5 * w * h).reshape(5, h, w)

    def keypoints(self, o):
        n = self.count(o)
        return _arr(self.L.orc_sift_keypoints(self.h, o), 4 * n).reshape(n, 4)

    def orientations(self, o):
        n = self.count(o)
        return _arr(self.L.orc_sift_orientations(self.h, o), n)

    def descriptors(self, o):
        n = self.count(o)
        return _arr(self.L.orc_sift_descriptors(self.h, o), 128 * n).reshape(n, 128)

    def taps(self, level):
        buf = (C.c_float * 64)()
        fr = self.L.orc_sift_taps(self.h, level, buf, 64)
        return np.array(buf[:2 * fr + 1], dtype=np.float32), fr

    def all_features(self):
        """Octave-major concatenation, as the reference returns it (sift.cpp:1889-1911)."""
        kps, ors, des, octs = [], [], [], []
        for o in range(3):
            n = self.count(o)
            kps.append(self.keypoints(o))
            octs.append(np.full(n, o, dtype=np.int32))
            ors.append(self.orientations(o) if n else np.zeros(0, np.float32))
            des.append(self.descriptors(o) if n else np.zeros((0, 128), np.float32))
        return np.concatenate(kps), np.concatenate(octs), np.concatenate(ors), np.concatenate(des)


def match_top2(q, db, dim=None, return_scores=False):
    """CPU statement of mif_match_top2 / mif_match_top2_scores (same contract)."""
    q = np.ascontiguousarray(q, np.float32)
    db = np.ascontiguousarray(db, np.float32)
    ld = q.shape[1]
    dim = ld if dim is None else dim
    nq, nd = len(q), len(db)
    idx = np.full((nq, 2), -1, np.int32)
    dist = np.full((nq, 2), np.inf, np.float32)
    score = np.full((nq, 2), np.inf, np.float32)
    if nq:
        lib().orc_match_top2_scores(q.ctypes.data, nq, db.ctypes.data if nd else None, nd, dim, ld,
                                    idx.ctypes.data, dist.ctypes.data, score.ctypes.data)
    return (idx, dist, score) if return_scores else (idx, dist)


class SurfOracle:
    """One CPU run of the restated reference SURF pipeline."""

    def __init__(self, width, height, library=None):
        self.L = library or lib()
        self.h = self.L.orc_surf_create(width, height)
        if not self.h:
            raise ValueError("bad image geometry")
        self.width, self.height = width, height

    def close(self):
        if self.h:
            self.L.orc_surf_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def run(self, pixels):
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        ch = 1 if px.ndim == 2 else px.shape[2]
        assert px.size == self.width * self.height * ch
        return self.L.orc_surf_run(self.h, px.ctypes.data, ch)

    def count(self):
        return self.L.orc_surf_count(self.h)

    def raw_count(self):
        return self.L.orc_surf_raw_count(self.h)

    def integral(self):
        return _arr(self.L.orc_surf_integral(self.h), self.width * self.height).reshape(self.height, self.width)

    def layer_info(self, i):
        v = [C.c_int() for _ in range(4)]
        self.L.orc_surf_layer_info(self.h, i, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)          # w, h, step, filter

    def layer(self, i):
        w, h, _, _ = self.layer_info(i)
        if w * h == 0:
            return np.zeros((h, w), np.float32), np.zeros((h, w), np.int32)
        r = _arr(self.L.orc_surf_layer_responses(self.h, i), w * h).reshape(h, w)
        l = np.ctypeslib.as_array(self.L.orc_surf_layer_laplacians(self.h, i), shape=(w * h,)).copy().reshape(h, w)
        return r, l

    def points(self):
        """[N, 70] float32 rows: x, y, scale, orientation, laplacian, 0, descriptor[64]."""
        n = self.count()
        out = np.zeros((n, 70), np.float32)
        if n:
            self.L.orc_surf_points(self.h, out.ctypes.data)
        return out


def resized_size(width, height, ratio_x, ratio_y):
    """(int)(width * ratio_x), (int)(height * ratio_y) in float arithmetic (SIFT/sift.cpp:2108-2109)."""
    return int(np.float32(width) * np.float32(ratio_x)), int(np.float32(height) * np.float32(ratio_y))


def resize_u8(img, ratio_x, ratio_y):
    """SIFT::test_kResize of the reference on an (h, w) or (h, w, 3) uint8 image."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    c = 1 if img.ndim == 2 else img.shape[2]
    wd, hd = resized_size(w, h, ratio_x, ratio_y)
    out = np.zeros((hd, wd) if c == 1 else (hd, wd, c), np.uint8)
    lib().orc_resize_u8(img.ctypes.data, w, h, c, ratio_x, ratio_y, out.ctypes.data)
    return out


def gray_resize(img, ratio_x, ratio_y):
    """SIFT::test_ColorToGray_Resize of the reference on an (h, w, 3) BGR uint8 image."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    wd, hd = resized_size(w, h, ratio_x, ratio_y)
    out = np.zeros((hd, wd), np.float32)
    lib().orc_gray_resize(img.ctypes.data, w, h, ratio_x, ratio_y, out.ctypes.data)
    return out
