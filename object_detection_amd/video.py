"""Frame-sequence pipeline (mif_video_*): batches of frames extracted with SIFT and matched frame against
previous frame on the GPU, what the -v mode of the reference's demo does per frame
(SIFT/main.cpp:103-153)."""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _lib

FrameSummary = namedtuple("FrameSummary", "status keypoints good_matches query_rows")


class VideoPipeline:
    def __init__(self, width, height, channels=3, batch=64, device=0):
        self._lib = _lib.load_library()
        self._h = C.c_void_p()
        _lib.check(self._lib.mif_video_create(device, width, height, channels, batch, C.byref(self._h)))
        self.width, self.height, self.channels, self.batch = width, height, channels, batch
        self._waiting = 0
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mif_video_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_matching(self, enable=True):
        """Extraction only when disabled (good_matches = -1 for the batches submitted from then on)."""
        _lib.check(self._lib.mif_video_set_matching(self._h, 1 if enable else 0))

    def submit_device(self, d_frames_ptr, nframes=None):
        """d_frames_ptr: device address of uint8 [nframes][height][width][channels]."""
        n = self.batch if nframes is None else nframes
        _lib.check(self._lib.mif_video_submit_device(self._h, C.c_void_p(d_frames_ptr), n))
        self._waiting += n

    def submit_host(self, frames):
        """frames: uint8 array [nframes, height, width(, channels)] in host memory."""
        px = np.ascontiguousarray(frames, dtype=np.uint8)
        n = px.shape[0]
        if px.size != n * self.width * self.height * self.channels:
            raise ValueError("frames do not match %dx%dx%d" % (self.width, self.height, self.channels))
        _lib.check(self._lib.mif_video_submit_host(self._h, px.ctypes.data, n))
        self._keep = (self._keep + [px])[-3:]          # the upload may still be reading the buffer
        self._waiting += n

    def submit_host_ptr(self, frames_ptr, nframes):
        """frames_ptr: host address of uint8 [nframes][height][width][channels], e.g. pinned memory (the
        upload is then asynchronous; the buffer may be rewritten once the call after next has returned)."""
        _lib.check(self._lib.mif_video_submit_host(self._h, C.c_void_p(frames_ptr), nframes))
        self._waiting += nframes

    def collect(self):
        """Waits for everything submitted; returns a FrameSummary per frame since the last collect."""
        cap = max(self._waiting, 1)
        buf = (_lib.VideoFrame * cap)()
        n = C.c_int()
        _lib.check(self._lib.mif_video_collect(self._h, buf, cap, C.byref(n)))
        self._waiting = 0
        self._keep = []
        return [FrameSummary(f.status, (f.keypoints[0], f.keypoints[1], f.keypoints[2]), f.good_matches, f.query_rows)
                for f in buf[:n.value]]
