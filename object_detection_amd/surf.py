"""Host-side mirror of the reference's SURF interface (SURF/coopcl_surf/ccl_surf/ccl_surf.h) over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib

# ccl_surf::Ipoint (ccl_surf.h:11-19): 280 bytes
IPOINT_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("scale", np.float32), ("orientation", np.float32),
                         ("laplacian", np.int32), ("clusterIndex", np.int32), ("descriptor", np.float32, (64,))])


class SurfExtractor:
    """One GPU SURF pipeline for images of a fixed geometry."""

    def __init__(self, width, height, channels=1, device=0, batch=1):
        self._lib = _lib.load_library()
        h = C.c_void_p()
        _lib.check(self._lib.mif_surf_create_batched(device, width, height, channels, batch, C.byref(h)))
        self._h = h
        self.width, self.height, self.channels, self.device = width, height, channels, device
        self.batch = batch
        self.capacity = self._lib.mif_surf_capacity(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mif_surf_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def run(self, pixels):
        """pixels: uint8 [h, w] or [h, w, 3] (BGR). Returns a structured array of Ipoints
        (empty when no interest point is found, like ccl_surf::run) and the raw count."""
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        if px.size != self.width * self.height * self.channels:
            raise ValueError("pixel buffer does not match %dx%dx%d" % (self.width, self.height, self.channels))
        pts = np.zeros(self.capacity, IPOINT_DTYPE)
        n, raw, ms = C.c_int(), C.c_int(), C.c_float()
        _lib.check(self._lib.mif_surf_run(self._h, px.ctypes.data, pts.ctypes.data, self.capacity, C.byref(n),
                                          C.byref(raw), C.byref(ms)))
        self.last_elapsed_ms = ms.value
        return pts[:n.value].copy(), raw.value

    def enqueue(self, d_pixels_ptr):
        _lib.check(self._lib.mif_surf_enqueue(self._h, C.c_void_p(d_pixels_ptr)))

    def enqueue_batch(self, d_pixels_ptr, nframes):
        """d_pixels_ptr: device address of nframes (<= batch) frames of uint8 pixels, back to back."""
        _lib.check(self._lib.mif_surf_enqueue_batch(self._h, C.c_void_p(d_pixels_ptr), nframes))

    def copy_points(self, frame=0):
        """Ipoints of one frame of the last batch (structured array) and the raw count."""
        pts = np.zeros(self.capacity, IPOINT_DTYPE)
        n, raw = C.c_int(), C.c_int()
        _lib.check(self._lib.mif_surf_copy_points(self._h, frame, pts.ctypes.data, self.capacity, C.byref(n),
                                                  C.byref(raw)))
        return pts[:n.value].copy(), raw.value

    def sync(self):
        _lib.check(self._lib.mif_surf_sync(self._h))

    def stream(self):
        return self._lib.mif_surf_stream(self._h)

    def device_results(self):
        v = [C.c_void_p() for _ in range(6)]
        _lib.check(self._lib.mif_surf_device_results(self._h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)      # pos, scale, laplacian, orientation, descriptors, counts

    STAGES = ("integral_image", "hessian_det", "non_max_supression", "getOrientation", "createDescriptors")

    def integral_fallbacks(self):
        """Batches whose integral images the launch-per-pass form had to build again (expected: 0)."""
        n = C.c_int(0)
        _lib.check(self._lib.mif_surf_integral_fallbacks(self._h, C.byref(n)))
        return n.value

    def set_timing(self, enable=True):
        _lib.check(self._lib.mif_surf_set_timing(self._h, 1 if enable else 0))

    def stage_times(self):
        """ms per stage (STAGES) of the last batch enqueued with timing on."""
        buf = (C.c_float * 5)()
        _lib.check(self._lib.mif_surf_stage_times(self._h, buf))
        return np.array(buf[:], dtype=np.float32)

    def integral(self, frame=0):
        out = np.empty((self.height, self.width), np.float32)
        _lib.check(self._lib.mif_surf_copy_integral_frame(self._h, frame, out.ctypes.data))
        return out

    def layer_info(self, i):
        v = [C.c_int() for _ in range(4)]
        _lib.check(self._lib.mif_surf_layer_info(self._h, i, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)      # w, h, step, filter

    def layer(self, i):
        w, h, _, _ = self.layer_info(i)
        r = np.zeros((h, w), np.float32)
        l = np.zeros((h, w), np.int32)
        _lib.check(self._lib.mif_surf_copy_layer(self._h, i, r.ctypes.data, l.ctypes.data))
        return r, l


_default = {}
_initialized = False


def initialize_surf_driver(kernel_path_name=(), dump=False, offload=0.0, jit_flags="", app_begin=None):
    """ccl_surf::initialize_surf_driver (ccl_surf.h:23-28). The kernel file list, JIT flags and the
    CPU/GPU offload fraction belong to the replaced OpenCL driver; they are accepted and ignored."""
    global _initialized
    if offload != -1.0 and not (0.0 <= offload <= 1.0):
        raise ValueError("offload must be -1 or within [0, 1]")
    _lib.load_library()
    _initialized = True
    return True


def run(image, use_images_in_kernelfunc=False, device=0):
    """ccl_surf::run (ccl_surf.h:30): returns the Ipoint array; empty if the driver was not
    initialised (ccl_surf.cpp:791-795) or no interest point exists. Keeps one pipeline per device and
    rebuilds it when the image grows (ccl_surf.cpp:806-810) -- here: whenever the geometry changes."""
    if not _initialized:
        return np.zeros(0, IPOINT_DTYPE)
    img = np.asarray(image)
    if img.dtype != np.uint8 or img.ndim not in (2, 3):
        raise ValueError("image must be uint8 [h,w] or [h,w,3]")
    h, w = img.shape[:2]
    c = 1 if img.ndim == 2 else img.shape[2]
    ex = _default.get(device)
    if ex is None or (ex.width, ex.height, ex.channels) != (w, h, c):
        if ex is not None:
            ex.close()
        ex = SurfExtractor(w, h, c, device=device)
        _default[device] = ex
    return ex.run(img)[0]
