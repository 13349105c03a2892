"""The C-ABI library loads and exports every symbol include/mi_features.h declares (no GPU needed)."""
import os
import re

import numpy as np
import pytest

import object_detection_amd as oda
from object_detection_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi_features.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mif_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    lib = oda.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 20
    bound = {name for name, _, _ in _lib.SYMBOLS}
    for name in declared:
        assert hasattr(lib, name), "libmi_features.so does not export " + name
        assert name in bound, "python binding lacks " + name


def test_argument_errors_need_no_gpu():
    lib = oda.load_library()
    import ctypes as C
    h = C.c_void_p()
    assert lib.mif_sift_create(0, 0, 10, 1, 0, C.byref(h)) == _lib.MIF_ERR_INVALID_ARGUMENT
    assert lib.mif_sift_create(0, 10, 10, 2, 0, C.byref(h)) == _lib.MIF_ERR_INVALID_ARGUMENT
    assert lib.mif_match_create(0, 0, 5, C.byref(h)) == _lib.MIF_ERR_INVALID_ARGUMENT


def test_product_fails_loudly_without_library(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "_HERE", str(tmp_path))
    with pytest.raises(ImportError):
        oda.load_library()


def test_offload_argument_is_range_checked():
    img = np.zeros((16, 16), np.uint8)
    with pytest.raises(ValueError):
        oda.extract_features(img, offload=1.5)
    with pytest.raises(ValueError):
        oda.extract_features(img, offload=-0.5)


def test_pack_sift_rows_shift():
    d = np.arange(256, dtype=np.float32).reshape(2, 128)
    p = oda.pack_sift_rows(d)
    assert p[0, 0] == 0 and p[1, 0] == 0
    assert np.array_equal(p[:, 1:], d[:, :127])
