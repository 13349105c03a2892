"""One small image through mif_surf_run many times (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import object_detection_amd as oda
from oracle_lib import read_pnm
img = read_pnm(os.path.join(ROOT, "tests", "golden", sys.argv[1] if len(sys.argv) > 1 else "box.pgm"))
ex = oda.SurfExtractor(img.shape[1], img.shape[0], 1)
for _ in range(5): ex.run(img)
t0 = time.perf_counter()
for _ in range(200): pts, raw = ex.run(img)
print("lib %s: %.4f ms per frame, %d points, gpu %.4f ms" % (os.environ.get("MIF_LIBRARY", "default"), (time.perf_counter() - t0) / 200 * 1e3, len(pts), ex.last_elapsed_ms))
