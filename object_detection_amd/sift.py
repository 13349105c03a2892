"""Host-side mirror of the reference's SIFT interface (SIFT/sift.h:42-71) over the C ABI."""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _lib

# keypoints [N,4] float32 {x, y, sigma, level}; octave_ids [N] int32; orientations [N]; descriptors [N,128]
SiftFeatures = namedtuple("SiftFeatures", "keypoints octave_ids orientations descriptors counts elapsed_ms")


class SiftExtractor:
    """One GPU pipeline for images of a fixed geometry (the reference rebuilds its
    global sift_algorithm when the size changes, SIFT/sift.cpp:2161-2179)."""

    def __init__(self, width, height, channels=1, device=0, up_scale=False, batch=1):
        self._lib = _lib.load_library()
        h = C.c_void_p()
        _lib.check(self._lib.mif_sift_create_batched(device, width, height, channels, 1 if up_scale else 0, batch,
                                                     C.byref(h)))
        self._h = h
        self.width, self.height, self.channels, self.device = width, height, channels, device
        self.batch = batch
        self.capacity = self._lib.mif_sift_capacity(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mif_sift_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def octave_info(self, octave):
        w, h, m = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._lib.mif_sift_octave_info(self._h, octave, C.byref(w), C.byref(h), C.byref(m)))
        return w.value, h.value, m.value

    def extract(self, pixels, want_descriptors=True):
        """pixels: uint8 array with height*width*channels elements (host). Returns
        (status, SiftFeatures); status is 0 or -1 (an octave without keypoints,
        SIFT/sift.cpp:1257-1258), in which case the feature arrays are empty."""
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        if px.size != self.width * self.height * self.channels:
            raise ValueError("pixel buffer does not match %dx%dx%d" % (self.width, self.height, self.channels))
        cap = max(self.capacity, 1)
        kp = np.empty((cap, 4), np.float32)
        ori = np.empty(cap, np.float32) if want_descriptors else None
        des = np.empty((cap, 128), np.float32) if want_descriptors else None
        counts = (C.c_int * 3)()
        ms = C.c_float()
        rc = self._lib.mif_sift_extract(
            self._h, px.ctypes.data, kp.ctypes.data,
            ori.ctypes.data if want_descriptors else None,
            des.ctypes.data if want_descriptors else None, cap, counts, C.byref(ms))
        _lib.check(rc, allow=(_lib.MIF_ERR_NO_KEYPOINTS,))
        cnt = [counts[0], counts[1], counts[2]]
        n = sum(cnt)
        octs = np.repeat(np.arange(3, dtype=np.int32), cnt)
        feats = SiftFeatures(kp[:n].copy(), octs,
                             ori[:n].copy() if want_descriptors else np.zeros(0, np.float32),
                             des[:n].copy() if want_descriptors else np.zeros((0, 128), np.float32),
                             cnt, ms.value)
        return rc, feats

    # --- device-resident path -------------------------------------------------
    def enqueue_pyramid(self, d_pixels_ptr, nframes=1):
        """Phase 1 of enqueue(): base images and scale spaces of all octaves."""
        _lib.check(self._lib.mif_sift_enqueue_pyramid(self._h, C.c_void_p(d_pixels_ptr), nframes))

    def enqueue_features(self, want_descriptors=True):
        """Phase 2 of enqueue(): detection, orientations, descriptors."""
        _lib.check(self._lib.mif_sift_enqueue_features(self._h, 1 if want_descriptors else 0))

    def snapshot_frame(self, frame, d_descriptors_ptr, d_count_ptr):
        _lib.check(self._lib.mif_sift_snapshot_frame(self._h, frame, C.c_void_p(d_descriptors_ptr),
                                                     C.c_void_p(d_count_ptr)))

    def pack_results(self, nframes, d_keypoints, d_orientations, d_descriptors, d_offsets, max_rows):
        """The rows of the first nframes frames of the last batch back to back in the caller's device
        arrays (integer addresses; orientations and descriptors may both be None), in stream order;
        d_offsets takes int32[nframes + 2] (see mif_sift_pack_results)."""
        _lib.check(self._lib.mif_sift_pack_results(
            self._h, nframes, C.c_void_p(d_keypoints), C.c_void_p(d_orientations) if d_orientations else None,
            C.c_void_p(d_descriptors) if d_descriptors else None, C.c_void_p(d_offsets), max_rows))

    def enqueue(self, d_pixels_ptr, nframes=1, want_descriptors=True):
        """d_pixels_ptr: integer device address of nframes frames of uint8 pixels, back to back."""
        _lib.check(self._lib.mif_sift_enqueue(self._h, C.c_void_p(d_pixels_ptr), nframes,
                                              1 if want_descriptors else 0))

    def wait_for(self, other):
        """Work enqueued on this handle from now on waits for what `other` has enqueued so far."""
        _lib.check(self._lib.mif_sift_wait_for(self._h, other._h))

    def copy_results(self, frame=0, want_descriptors=True):
        """Host copy of one frame of the last enqueued batch -> (status, SiftFeatures)."""
        cap = max(self.capacity, 1)
        kp = np.empty((cap, 4), np.float32)
        ori = np.empty(cap, np.float32) if want_descriptors else None
        des = np.empty((cap, 128), np.float32) if want_descriptors else None
        counts = (C.c_int * 3)()
        rc = self._lib.mif_sift_copy_results(self._h, frame, kp.ctypes.data,
                                             ori.ctypes.data if want_descriptors else None,
                                             des.ctypes.data if want_descriptors else None, cap, counts)
        _lib.check(rc, allow=(_lib.MIF_ERR_NO_KEYPOINTS,))
        cnt = [counts[0], counts[1], counts[2]]
        n = sum(cnt)
        octs = np.repeat(np.arange(3, dtype=np.int32), cnt)
        return rc, SiftFeatures(kp[:n].copy(), octs,
                                ori[:n].copy() if want_descriptors else np.zeros(0, np.float32),
                                des[:n].copy() if want_descriptors else np.zeros((0, 128), np.float32), cnt, 0.0)

    def sync(self):
        _lib.check(self._lib.mif_sift_sync(self._h))

    def stream(self):
        return self._lib.mif_sift_stream(self._h)

    def wait_input_read(self, waiter_stream):
        """Work enqueued on `waiter_stream` from now on waits until the last batch's pixels have been read."""
        _lib.check(self._lib.mif_sift_wait_input_read(self._h, C.c_void_p(waiter_stream)))

    def set_stream(self, stream):
        """Enqueue on the caller's hipStream_t from now on (None: the handle's own stream again)."""
        _lib.check(self._lib.mif_sift_set_stream(self._h, C.c_void_p(stream) if stream else None))

    def device_results(self):
        kp, ori, des, cnt = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(self._lib.mif_sift_device_results(self._h, C.byref(kp), C.byref(ori), C.byref(des), C.byref(cnt)))
        return kp.value, ori.value, des.value, cnt.value

    def read_counts(self, frame=0):
        c, r = (C.c_int * 3)(), (C.c_int * 3)()
        status = self._lib.mif_sift_read_counts(self._h, frame, c, r)
        if status not in (0, _lib.MIF_ERR_NO_KEYPOINTS):
            _lib.check(status)
        return status, list(c), list(r)

    def set_timing(self, enable=True):
        _lib.check(self._lib.mif_sift_set_timing(self._h, 1 if enable else 0))

    def stage_times(self):
        """ms per [octave][stage] of the last batch enqueued with timing on; stages: base image,
        scale space (6 blur launches), detector, orientation + descriptor."""
        buf = (C.c_float * 12)()
        _lib.check(self._lib.mif_sift_stage_times(self._h, buf))
        return np.array(buf[:], dtype=np.float32).reshape(3, 4)

    def blur_launches(self):
        """Scale-space kernel launches of the last enqueued batch (18 with one launch per level)."""
        n = self._lib.mif_sift_blur_launches(self._h)
        if n < 0:
            _lib.check(n)
        return n

    def last_elapsed_ms(self):
        ms = C.c_float()
        _lib.check(self._lib.mif_sift_last_elapsed_ms(self._h, C.byref(ms)))
        return ms.value

    # --- probes (SIFT/sift.h:13-21 get_blur / get_diff_blur) --------------------
    def level(self, octave, level, frame=0):
        w, h, _ = self.octave_info(octave)
        out = np.empty((h, w), np.float32)
        _lib.check(self._lib.mif_sift_copy_level(self._h, frame, octave, level, out.ctypes.data))
        return out

    def dog(self, octave, index, frame=0):
        w, h, _ = self.octave_info(octave)
        out = np.empty((h, w), np.float32)
        _lib.check(self._lib.mif_sift_copy_dog(self._h, frame, octave, index, out.ctypes.data))
        return out

    def base(self, octave, frame=0):
        """Input of the octave's scale space (gray conversion resp. decimated previous octave)."""
        w, h, _ = self.octave_info(octave)
        out = np.empty((h, w), np.float32)
        _lib.check(self._lib.mif_sift_copy_base(self._h, frame, octave, out.ctypes.data))
        return out

    def taps(self, level):
        buf = (C.c_float * 64)()
        fr = self._lib.mif_sift_taps(self._h, level, buf, 64)
        return np.array(buf[:2 * fr + 1], dtype=np.float32), fr


_default = {}


def extract_features(image, offload=1.0, want_descriptors=True, device=0, up_scale=False):
    """Python spelling of SIFT::extract_features (SIFT/sift.h:51-69): image is a
    uint8 array [h, w] or [h, w, 3] (BGR). `offload` is the reference's CPU/GPU
    split; it is range-checked like common/parse_args.h:89-95 and otherwise
    ignored (everything runs on the GPU). Keeps one extractor per device and
    rebuilds it when the image size changes, like the reference's global."""
    if offload != -1.0 and not (0.0 <= offload <= 1.0):
        raise ValueError("offload must be -1 or within [0, 1]")
    img = np.asarray(image)
    if img.dtype != np.uint8 or img.ndim not in (2, 3):
        raise ValueError("image must be uint8 [h,w] or [h,w,3]")
    h, w = img.shape[:2]
    c = 1 if img.ndim == 2 else img.shape[2]
    ex = _default.get(device)
    if ex is None or (ex.width, ex.height, ex.channels) != (w, h, c):
        if ex is not None:
            ex.close()
        ex = SiftExtractor(w, h, c, device=device, up_scale=up_scale)
        _default[device] = ex
    return ex.extract(img, want_descriptors)


def test_kResize(image, ratio_x, ratio_y, device=0):
    """SIFT::test_kResize (SIFT/sift.h:28-32): nearest-neighbour resampling of an (h, w) or (h, w, 3) uint8
    image by (ratio_x, ratio_y) on the GPU."""
    image = np.ascontiguousarray(image, np.uint8)
    h, w = image.shape[:2]
    c = 1 if image.ndim == 2 else image.shape[2]
    wd, hd = int(np.float32(w) * np.float32(ratio_x)), int(np.float32(h) * np.float32(ratio_y))
    out = np.zeros((max(hd, 0), max(wd, 0)) if c == 1 else (max(hd, 0), max(wd, 0), c), np.uint8)
    _lib.check(_lib.load_library().mif_probe_resize_u8(device, image.ctypes.data, w, h, c, ratio_x, ratio_y,
                                                       out.ctypes.data))
    return out


def test_ColorToGray_Resize(image, ratio_x, ratio_y, device=0):
    """SIFT::test_ColorToGray_Resize (SIFT/sift.h:34-38): gray conversion of an (h, w, 3) BGR uint8 image,
    then nearest-neighbour resampling by (ratio_x, ratio_y), on the GPU."""
    image = np.ascontiguousarray(image, np.uint8)
    h, w = image.shape[:2]
    c = 1 if image.ndim == 2 else image.shape[2]
    wd, hd = int(np.float32(w) * np.float32(ratio_x)), int(np.float32(h) * np.float32(ratio_y))
    out = np.zeros((max(hd, 0), max(wd, 0)), np.float32)
    _lib.check(_lib.load_library().mif_probe_gray_resize(device, image.ctypes.data, w, h, c, ratio_x, ratio_y,
                                                         out.ctypes.data))
    return out


test_kResize.__test__ = False                 # not test cases of this package's own test suite
test_ColorToGray_Resize.__test__ = False
