"""mif_match_top2_device_noalloc (the entry point that allocates nothing: its workspace is a fixed array
in device memory) against mif_match_top2_device, through the C ABI: indices and distance bits equal, on
two calls in a row, the second with a smaller input, then at the limits of the fixed workspace."""
import ctypes as C

import numpy as np
import pytest

import object_detection_amd as oda
from object_detection_amd import _lib
from synth import sift_like_descriptors

pytestmark = pytest.mark.gpu


def _both(torch, lib, m, q, db, dim, stream=None):
    dev = torch.device("cuda:0")
    nq, nd, ld = len(q), len(db), q.shape[1]
    tq = torch.from_numpy(q).to(dev)
    td = torch.from_numpy(db).to(dev) if nd else torch.zeros((1, ld), dtype=torch.float32, device=dev)
    out = []
    for which in (0, 1):
        idx = torch.full((max(nq, 1), 2), -7, dtype=torch.int32, device=dev)
        dist = torch.full((max(nq, 1), 2), -7.0, dtype=torch.float32, device=dev)
        if which == 0:
            m.top2_device(tq.data_ptr(), nq, td.data_ptr(), nd, dim, ld, idx.data_ptr(), dist.data_ptr(), stream=stream)
        else:
            _lib.check(lib.mif_match_top2_device_noalloc(0, C.c_void_p(tq.data_ptr()), nq, C.c_void_p(td.data_ptr()), nd,
                                                         dim, ld, C.c_void_p(idx.data_ptr()), C.c_void_p(dist.data_ptr()),
                                                         C.c_void_p(stream) if stream else None))
        torch.cuda.synchronize()
        out.append((idx.cpu().numpy(), dist.cpu().numpy()))
    (i0, d0), (i1, d1) = out
    assert np.array_equal(i0, i1)
    assert np.array_equal(d0.view(np.uint32), d1.view(np.uint32))
    return i1, d1


def test_noalloc_entry_point_equals_matcher_two_calls_in_a_row():
    import torch
    lib = oda.load_library()
    m = oda.Matcher(4096, 65536)
    m.set_screening(False)
    # first call: a large input that uses many database splits
    db = sift_like_descriptors(30000, seed=41)
    q = sift_like_descriptors(1500, seed=42)
    db[123] = q[7]
    db[20000] = db[29000] = q[9]                        # a tie across splits: the lower index wins
    i, d = _both(torch, lib, m, q, db, 128)
    assert i[7, 0] == 123 and d[7, 0] == 0 and list(i[9]) == [20000, 29000]
    # second call straight after it, smaller in every respect (fewer rows, fewer splits, dim 127 of 128):
    # nothing of the first call's workspace contents may leak into it
    i, d = _both(torch, lib, m, q[:37], db[:211], 127)
    assert i.shape == (37, 2) and np.all(i >= 0)
    # and on a stream of the caller's
    s = torch.cuda.Stream()
    _both(torch, lib, m, q[:300], db[:5000], 128, stream=s.cuda_stream)
    # one database row, then none
    i, d = _both(torch, lib, m, q[:5], db[:1], 128)
    assert np.all(i[:, 1] == -1) and np.all(np.isinf(d[:, 1]))
    i, d = _both(torch, lib, m, q[:5], db[:0], 128)
    assert np.all(i == -1)
    # SURF-sized rows (dimension 64, rows of 64 floats)
    rng = np.random.RandomState(3)
    s1 = rng.standard_normal((200, 64)).astype(np.float32)
    s2 = rng.standard_normal((777, 64)).astype(np.float32)
    _both(torch, lib, m, s1, s2, 64)


def test_noalloc_limits_and_errors():
    import torch
    lib = oda.load_library()
    dev = torch.device("cuda:0")
    m = oda.Matcher(4096, 65536)
    m.set_screening(False)
    # the largest call the fixed workspace takes
    db = np.tile(sift_like_descriptors(8192, seed=51), (8, 1))
    db += np.arange(len(db), dtype=np.float32)[:, None] * np.float32(1e-3)
    q = sift_like_descriptors(4096, seed=52)
    _both(torch, lib, m, q, np.ascontiguousarray(db), 128)
    buf = torch.zeros((8, 128), dtype=torch.float32, device=dev)
    out_i = torch.zeros((8, 2), dtype=torch.int32, device=dev)
    out_d = torch.zeros((8, 2), dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    assert lib.mif_match_top2_device_noalloc(0, p(buf), 4097, p(buf), 8, 128, 128, p(out_i), p(out_d), None) == _lib.MIF_ERR_CAPACITY
    assert b"fixed workspace" in lib.mif_last_error()
    assert lib.mif_match_top2_device_noalloc(0, p(buf), 8, p(buf), 65537, 128, 128, p(out_i), p(out_d), None) == _lib.MIF_ERR_CAPACITY
    assert lib.mif_match_top2_device_noalloc(0, p(buf), 8, p(buf), 8, 129, 129, p(out_i), p(out_d), None) == _lib.MIF_ERR_INVALID_ARGUMENT
    assert lib.mif_match_top2_device_noalloc(0, None, 8, p(buf), 8, 128, 128, p(out_i), p(out_d), None) == _lib.MIF_ERR_INVALID_ARGUMENT
    assert lib.mif_match_top2_device_noalloc(99, p(buf), 8, p(buf), 8, 128, 128, p(out_i), p(out_d), None) == _lib.MIF_ERR_INVALID_ARGUMENT
    assert lib.mif_match_top2_device_noalloc(0, p(buf), 0, p(buf), 8, 128, 128, p(out_i), p(out_d), None) == 0
