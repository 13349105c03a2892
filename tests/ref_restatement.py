"""Second, independent restatement of the reference's kOrientation, kDescriptor and kNormalize kernels
(SIFT/sift.cpp:431-582, 585-709, 712-765) in plain Python over numpy float32 / float64 scalars.

TEST INFRASTRUCTURE ONLY. It exists to cross-check oracle/sift_oracle.cpp: both were written from the
reference's kernel text, separately, and must agree bit for bit on the same keypoints. Scalar loops, so
only a handful of keypoints per test. The elementary functions (exp, atan2, sin, cos) are the shared ones
of csrc/mif_math.h through the oracle library; sqrt and division are IEEE in numpy as in C.
"""
import ctypes as C
import math

import numpy as np

import oracle_lib as orc

f32 = np.float32
f64 = np.float64
M_PI = f64(3.14159265358979323846)          # sift.cpp:48
M_PI2 = f64(2.0) * M_PI                     # sift.cpp:50
RPI = f64(4.0) / M_PI                       # sift.cpp:51
ORI_BINS = 36


def _m():
    L = orc.lib()
    for n in ("orc_math_expf", "orc_math_sinf", "orc_math_cosf"):
        getattr(L, n).argtypes = [C.c_float]
        getattr(L, n).restype = C.c_float
    L.orc_math_atan2f.argtypes = [C.c_float, C.c_float]
    L.orc_math_atan2f.restype = C.c_float
    L.orc_math_atan_pos.argtypes = [C.c_float]
    L.orc_math_atan_pos.restype = C.c_float
    return L


def _round_half_away(x):
    """OpenCL round() on a double."""
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def orientation(pyr, w, h, kp, oid):
    """pyr: flat float32 pyramid [6*h*w] of the octave; kp = (x, y, sigma, level) float32. Returns float32."""
    L = _m()
    fx, fy, fz, fw = (f32(v) for v in kp)
    hist = np.zeros((16, ORI_BINS), np.float32)
    exp_denom = f32(f64(fz) * f64(1.5))
    rad = int(_round_half_away(float(f64(4.5) * f64(fz))))
    exp_factor = f32(-0.5) / (exp_denom * exp_denom)
    sq_thres = rad * rad
    offset_py = int(f32(w * h) * fw)
    p2 = f32(2.0 ** oid)
    x = int(fx / p2)
    y = int(fy / p2)
    xmin, ymin = max(1, x - rad), max(1, y - rad)
    xmax, ymax = min(w - 2, x + rad), min(h - 2, y + rad)
    wx, hy = xmax - xmin + 1, ymax - ymin + 1
    loops = wx * hy
    for lid in range(16):
        for i in range(lid, loops, 16):
            yy = i // wx + ymin
            xx = i % wx + xmin
            if 0 < xx < w - 1 and 0 < yy < h - 1:
                dx = pyr[yy * w + xx + 1 + offset_py] - pyr[yy * w + xx - 1 + offset_py]
                dy = pyr[(yy - 1) * w + xx + offset_py] - pyr[(yy + 1) * w + xx + offset_py]
                grad = np.sqrt(f32(dx * dx + dy * dy))
                theta = f32(L.orc_math_atan2f(float(dy), float(dx)))
                ddx, ddy = f32(xx - x), f32(yy - y)
                sq_dist = int(ddx * ddx + ddy * ddy)
                if sq_dist > sq_thres:
                    continue
                b = int(_round_half_away(float((f64(ORI_BINS) * (f64(theta) + M_PI)) / M_PI2)))
                b = b if b < ORI_BINS else 0
                hist[lid, b] = hist[lid, b] + grad * f32(L.orc_math_expf(float(f32(sq_dist) * exp_factor)))
    for i in range(ORI_BINS):                   # the reference's 8-4-2-1 tree
        for half in (8, 4, 2, 1):
            for lid in range(half):
                hist[lid, i] = hist[lid, i] + hist[lid + half, i]
    hs = hist[0]
    for _ in range(2):
        h0, hp = hs[0], hs[ORI_BINS - 1]
        for i in range(ORI_BINS):
            hc = hs[i]
            hn = h0 if i + 1 == ORI_BINS else hs[i + 1]
            hs[i] = f32((f64(0.25) * f64(hp) + f64(0.5) * f64(hc)) + f64(0.25) * f64(hn))
            hp = hc
    v = hs[0]
    for i in range(1, ORI_BINS):
        if hs[i] > v:
            v = hs[i]
    v = f32(f64(0.8) * f64(v))
    result = f32(-1.0)
    for i in range(ORI_BINS):
        hc, hp, hn = hs[i], hs[ORI_BINS - 1 if i == 0 else i - 1], hs[(i + 1) % ORI_BINS]
        if hc >= v and hc > hn and hc > hp:
            fbin = f32(f64(f32(i)) + f64(f32(0.5) * (hp - hn)) / ((f64(hp) - f64(2.0) * f64(hc)) + f64(hn)))
            if fbin < 0.0:
                fbin = fbin + f32(ORI_BINS)
            elif fbin >= f32(ORI_BINS):
                fbin = fbin - f32(ORI_BINS)
            result = f32(((M_PI2 * f64(fbin)) / f64(ORI_BINS)) - M_PI)
    return result


def descriptor(pyr, w, h, kp, orient, oid):
    """Un-normalised 128 values of kDescriptor followed by kNormalize. Returns float32[128]."""
    L = _m()
    fx, fy, fz, fw = (f32(v) for v in kp)
    des = np.zeros((16, 8, 9), np.float32)
    p2 = f32(2.0 ** oid)
    x = int(fx / p2)
    y = int(fy / p2)
    offset_py = int(f32(w * h) * fw)
    ang = -f32(orient)
    spt = f32(abs(f64(fz) * f64(3.0)))
    s = f32(L.orc_math_sinf(float(ang)))
    c = f32(L.orc_math_cosf(float(ang)))
    cspt, sspt = c * spt, s * spt
    crspt, srspt = c / spt, s / spt
    for bid in range(16):
        ix, iy = bid & 3, bid >> 2
        offx, offy = f32(ix) - f32(1.5), f32(iy) - f32(1.5)
        ptx = (cspt * offx - sspt * offy) + f32(x)
        pty = (cspt * offy + sspt * offx) + f32(y)
        bsz = abs(cspt) + abs(sspt)

        def maxv(a, b):      # ((A>=B)?(A):(B)) with an int and a float: compared and returned in float
            return f32(a) if f32(a) >= b else b

        def minv(a, b):
            return f32(a) if f32(a) <= b else b
        xmin = int(maxv(1, np.floor(ptx - bsz)))
        ymin = int(maxv(1, np.floor(pty - bsz)))
        xmax = int(minv(w - 2, np.floor(ptx + bsz)))
        ymax = int(minv(h - 2, np.floor(pty + bsz)))
        wx, hy = xmax - xmin + 1, ymax - ymin + 1
        loops = wx * hy
        if wx <= 0 or hy <= 0:
            continue          # (a window empty in both directions would index outside the image)
        for lid in range(8):
            d = des[bid, lid]
            for i in range(lid, loops, 8):
                yy = i // wx + ymin
                xx = i % wx + xmin
                dx = f32(xx) - ptx
                dy = f32(yy) - pty
                nx = crspt * dx + srspt * dy
                ny = crspt * dy - srspt * dx
                nxn, nyn = abs(nx), abs(ny)
                if nxn < 1.0 and nyn < 1.0:
                    gx = pyr[yy * w + xx + 1 + offset_py] - pyr[yy * w + xx - 1 + offset_py]
                    gy = pyr[(yy + 1) * w + xx + offset_py] - pyr[(yy - 1) * w + xx + offset_py]
                    mod = np.sqrt(f32(gx * gx + gy * gy))
                    th = f32(L.orc_math_atan2f(float(gy), float(gx)))
                    dnx, dny = nx + offx, ny + offy
                    ww = f32(L.orc_math_expf(float(f32(-0.125) * (dnx * dnx + dny * dny))))
                    wwx = f32(f64(1.0) - f64(nxn))
                    wwy = f32(f64(1.0) - f64(nyn))
                    wgt = ((ww * wwx) * wwy) * mod
                    th = th - ang
                    while th < 0.0:
                        th = f32(f64(th) + M_PI2)
                    while f64(th) >= M_PI2:
                        th = f32(f64(th) - M_PI2)
                    tth = f32(f64(th) * RPI)
                    fo0 = int(np.floor(tth))
                    do0 = tth - f32(fo0)
                    fo = fo0 % 8
                    d[fo] = d[fo] + (f32(1.0) - do0) * wgt
                    d[fo + 1] = d[fo + 1] + do0 * wgt
            d[0] = d[0] + d[8]
    out = np.zeros(128, np.float32)
    for t in range(128):
        lid, bid = t & 7, t >> 3
        b = f32(0.0)
        for i in range(8):
            b = b + des[bid, i, lid]
        out[t] = b
    # kNormalize
    n1 = f32(0.0)
    for t in range(128):
        n1 = n1 + out[t] * out[t]
    n1 = f32(1.0) / np.sqrt(n1)
    v = np.minimum(np.maximum(out * n1, f32(0.0)), f32(0.2)).astype(np.float32)
    n2 = f32(0.0)
    for t in range(128):
        n2 = n2 + v[t] * v[t]
    n2 = f32(1.0) / np.sqrt(n2)
    return np.minimum(np.maximum(f32(512.0) * (v * n2), f32(0.0)), f32(255.0)).astype(np.float32)


def detector(dog, w, h, oid, max_feat):
    """kDetector (SIFT/sift.cpp:265-417) for the three launches s = 1, 2, 3 of one octave, in raster
    order of the detecting pixel (the order the checker and the GPU path emit; the reference's order is
    that of an atomic counter). dog: float32 [5, h, w]. Returns float32 [n, 4] = (x, y, sigma, level)."""
    L = orc.lib()
    L.orc_math_exp2f.argtypes = [C.c_float]
    L.orc_math_exp2f.restype = C.c_float
    B = 5
    out = []
    thr1 = f64(0.0108333)                       # _SIFT_CONTR_THR_1 as the JIT text carries it (6 digits)
    contr = f32(f64(0.065) / f64(3))            # (float)(_SIFT_CONTR_THR/_SIFT_INTVLS)
    p2 = f32(2.0 ** oid)
    for s in (1, 2, 3):
        P, Cc, N = dog[s - 1], dog[s], dog[s + 1]
        for y in range(B + 1, h - B):
            for x in range(B + 1, w - B):
                cur = Cc[y, x]
                if not (f64(abs(cur)) > thr1):
                    continue
                # the 26 comparisons exactly as the macro lists them: the previous image is read at
                # (y-1, x-1) twice and never at (y-1, x+1); the current image without its centre
                nb = [P[y, x - 1], P[y, x], P[y, x + 1], P[y + 1, x - 1], P[y + 1, x], P[y + 1, x + 1],
                      P[y - 1, x - 1], P[y - 1, x], P[y - 1, x - 1],
                      Cc[y, x - 1], Cc[y, x + 1], Cc[y + 1, x - 1], Cc[y + 1, x], Cc[y + 1, x + 1],
                      Cc[y - 1, x - 1], Cc[y - 1, x], Cc[y - 1, x + 1],
                      N[y, x - 1], N[y, x], N[y, x + 1], N[y + 1, x - 1], N[y + 1, x], N[y + 1, x + 1],
                      N[y - 1, x - 1], N[y - 1, x], N[y - 1, x + 1]]
                if not (all(cur > v for v in nb) or all(cur < v for v in nb)):
                    continue
                ty, tx, ts = y, x, s
                ok = False
                for it in range(5):
                    c0 = Cc[ty, tx]
                    gx = f32(0.5) * (Cc[ty, tx + 1] - Cc[ty, tx - 1])
                    gy = f32(0.5) * (Cc[ty + 1, tx] - Cc[ty - 1, tx])
                    gz = f32(0.5) * (N[ty, tx] - P[ty, tx])
                    Dxx = (Cc[ty, tx + 1] - f32(2.0) * c0) + Cc[ty, tx - 1]
                    Dyy = (Cc[ty + 1, tx] - f32(2.0) * c0) + Cc[ty - 1, tx]
                    Dss = (N[ty, tx] - f32(2.0) * c0) + P[ty, tx]
                    Dxy = f32(0.25) * (((Cc[ty + 1, tx + 1] + Cc[ty - 1, tx - 1]) - Cc[ty + 1, tx - 1]) - Cc[ty - 1, tx + 1])
                    Dxs = f32(0.25) * (((N[ty, tx + 1] + P[ty, tx - 1]) - N[ty, tx - 1]) - P[ty, tx + 1])
                    Dys = f32(0.25) * (((N[ty + 1, tx] + P[ty - 1, tx]) - P[ty + 1, tx]) - N[ty - 1, tx])
                    h0x = Dyy * Dss - Dys * Dys
                    h1x = Dys * Dxs - Dxy * Dss
                    h2x = Dxy * Dys - Dyy * Dxs
                    det = (Dxx * h0x + Dxy * h1x) + Dxs * h2x
                    h0y = Dxs * Dys - Dxy * Dss
                    h0z = Dxy * Dys - Dxs * Dyy
                    h1y = Dxx * Dss - Dxs * Dxs
                    h1z = Dxs * Dxy - Dxx * Dys
                    h2y = Dxy * Dxs - Dxx * Dys
                    h2z = Dxx * Dyy - Dxy * Dxy
                    if det == 0.0:
                        break
                    dv = f32(-1.0) / det
                    ox = ((h0x * dv) * gx + (h0y * dv) * gy) + (h0z * dv) * gz
                    oy = ((h1x * dv) * gx + (h1y * dv) * gy) + (h1z * dv) * gz
                    oz = ((h2x * dv) * gx + (h2y * dv) * gy) + (h2z * dv) * gz
                    if abs(ox) < 0.5 and abs(oy) < 0.5 and abs(oz) < 0.5:
                        ok = True
                        break
                    tx = int(f32(tx) + f32(_round_half_away(float(ox))))
                    if tx < B or tx >= w - B:
                        break
                    ty = int(f32(ty) + f32(_round_half_away(float(oy))))
                    if ty < B or ty >= h - B:
                        break
                    ts = int(f32(ts) + f32(_round_half_away(float(oz))))
                    if ts < 1 or ts > 3:
                        break
                if not ok:
                    continue
                trH = Dxx + Dyy
                detH = Dxx * Dyy - Dxy * Dxy
                if f64(abs(detH)) <= f64(1e-8):
                    continue
                if (trH * trH) / detH >= f32((10 + 1) * (10 + 1) // 10):
                    continue
                cp = P[ty, tx] if ts == 1 else (Cc[ty, tx] if ts == 2 else N[ty, tx])
                val = cp + f32(0.5) * ((gx * ox + gy * oy) + gz * oz)
                if abs(val) > contr:
                    sig = f32(f64(1.6) * f64(f32(L.orc_math_exp2f(float((f32(ts) + oz) / f32(3))))))
                    out.append(((f32(tx) + ox) * p2, (f32(ty) + oy) * p2, sig, f32(ts)))
    return np.array(out[:max_feat], np.float32).reshape(-1, 4)


# ------------------------------------------------------------------------------------------------ SURF
SURF_FILTER_MAP = [[0, 1, 2, 3], [1, 3, 4, 5], [3, 5, 6, 7]]      # ccl_surf.cpp:109-113, OCTAVES = 3
SURF_THRES = f32(0.0001)                                          # surf_constants.h


def surf_select_points(layers, max_points=8000):
    """selectIpoints + non_max_supression (ccl_surf.cpp:285-343, nonMaxSuppression_kernel.cl:79-364) in
    plain Python, points in (launch, row, column) order (the reference's order is that of an atomic
    counter). layers: list of (responses [h, w] float32, laplacians [h, w] int32, step, filter).
    Returns a list of (x, y, scale, laplacian)."""
    out = []

    def resp(layer, c, r, scale):                 # getResponse: layer[r*scale*width + c*scale]
        return layer[0][r * scale, c * scale]

    for o in range(3):
        for i in range(2):
            b = layers[SURF_FILTER_MAP[o][i]]
            m = layers[SURF_FILTER_MAP[o][i + 1]]
            t = layers[SURF_FILTER_MAP[o][i + 2]]
            tH, tW = t[0].shape
            mH, mW = m[0].shape
            bW = b[0].shape[1]
            tStep, tFilter, mFilter, bFilter = t[2], t[3], m[3], b[3]
            border = (tFilter + 1) // (2 * tStep)
            mS, bS = mW // tW, bW // tW
            # the launch covers the middle layer's size rounded up to 16; everything outside the top
            # layer's interior is rejected first
            for r in range(-(-mH // 16) * 16):
                if r <= border or r >= tH - border:
                    continue
                for c in range(-(-mW // 16) * 16):
                    if c <= border or c >= tW - border:
                        continue
                    cand = resp(m, c, r, mS)
                    if cand < SURF_THRES:
                        continue
                    lm = resp(t, c - 1, r - 1, 1)
                    for (dc, dr) in ((0, -1), (1, -1), (-1, 0), (0, 0), (1, 0), (-1, 1), (0, 1), (1, 1)):
                        lm = max(lm, resp(t, c + dc, r + dr, 1))
                    for (dc, dr) in ((-1, -1), (0, -1), (1, -1), (-1, 0), (0, 0), (1, 0), (-1, 1), (0, 1), (1, 1)):
                        lm = max(lm, resp(b, c + dc, r + dr, bS))
                    for (dc, dr) in ((-1, -1), (0, -1), (1, -1), (-1, 0), (1, 0), (-1, 1), (0, 1), (1, 1)):
                        lm = max(lm, resp(m, c + dc, r + dr, mS))
                    if lm > cand:
                        continue
                    # interpolateExtremum
                    two, four = f32(2.0), f32(4.0)
                    dx = (resp(m, c + 1, r, mS) - resp(m, c - 1, r, mS)) / two
                    dy = (resp(m, c, r + 1, mS) - resp(m, c, r - 1, mS)) / two
                    ds = (resp(t, c, r, 1) - resp(b, c, r, bS)) / two
                    v = resp(m, c, r, mS)
                    dxx = (resp(m, c + 1, r, mS) + resp(m, c - 1, r, mS)) - two * v
                    dyy = (resp(m, c, r + 1, mS) + resp(m, c, r - 1, mS)) - two * v
                    dss = (resp(t, c, r, 1) + resp(b, c, r, bS)) - two * v
                    dxy = (((resp(m, c + 1, r + 1, mS) - resp(m, c - 1, r + 1, mS)) - resp(m, c + 1, r - 1, mS))
                           + resp(m, c - 1, r - 1, mS)) / four
                    dxs = (((resp(t, c + 1, r, 1) - resp(t, c - 1, r, 1)) - resp(b, c + 1, r, bS))
                           + resp(b, c - 1, r, bS)) / four
                    dys = (((resp(t, c, r + 1, 1) - resp(t, c, r - 1, 1)) - resp(b, c, r + 1, bS))
                           + resp(b, c, r - 1, bS)) / four
                    H0, H1, H2, H3, H4, H5, H6, H7, H8 = dxx, dxy, dxs, dxy, dyy, dys, dxs, dys, dss
                    det = (H0 * (H4 * H8 - H7 * H5) - H1 * (H3 * H8 - H5 * H6)) + H2 * (H3 * H7 - H4 * H6)
                    inv = f32(1.0) / det
                    i0 = (H4 * H8 - H7 * H5) * inv
                    i1 = -(H3 * H8 - H5 * H6) * inv
                    i2 = (H3 * H7 - H6 * H4) * inv
                    i3 = -(H1 * H8 - H2 * H7) * inv
                    i4 = (H0 * H8 - H2 * H6) * inv
                    i5 = -(H0 * H7 - H6 * H1) * inv
                    i6 = (H1 * H5 - H2 * H4) * inv
                    i7 = -(H0 * H5 - H3 * H2) * inv
                    i8 = (H0 * H4 - H3 * H1) * inv
                    neg = f32(-1.0)
                    xc = i0 * dx * neg
                    xc = xc + i1 * dy * neg
                    xc = xc + i2 * ds * neg
                    xr = i3 * dx * neg
                    xr = xr + i4 * dy * neg
                    xr = xr + i5 * ds * neg
                    xi = i6 * dx * neg
                    xi = xi + i7 * dy * neg
                    xi = xi + i8 * ds * neg
                    if abs(xi) < 0.5 and abs(xr) < 0.5 and abs(xc) < 0.5:
                        x = (f32(c) + xc) * f32(tStep)
                        y = (f32(r) + xr) * f32(tStep)
                        sc = f32(0.1333) * (f32(mFilter) + xi * f32(mFilter - bFilter))
                        lap = int(m[1][r * mS, c * mS])
                        out.append((x, y, sc, lap))
    return out[:max_points]


# ------------------------------------------------------------------ SURF orientation and descriptor
#
# getOrientation_kernels.cl:68-316, createDescriptors_kernel.cl:135-275 and
# normalizeDescriptors_kernel.cl:54-111, one interest point at a time, every operation rounded to
# float32 where the kernel computes in float. Written from the kernel text, not from
# oracle/surf_oracle.cpp. Conventions this build fixes (the reference leaves them open): the 109
# samples of Step1 are ordered by work-item id instead of by atom_add arrival; dot(v, v) is summed
# left to right; exp / sin / cos / atan are the build's shared elementary functions.

SURF_PI = f32(3.141592654)                   # "#define pi 3.141592654f" (getOrientation_kernels.cl:57)
SURF_ID = [6, 5, 4, 3, 2, 1, 0, 1, 2, 3, 4, 5, 6]            # ccl_surf.cpp:753
SURF_GAUSS25 = [f32(v) for v in (             # ccl_surf.cpp:755-762
    0.02350693969273, 0.01849121369071, 0.01239503121241, 0.00708015417522, 0.00344628101733, 0.00142945847484, 0.00050524879060,
    0.02169964028389, 0.01706954162243, 0.01144205592615, 0.00653580605408, 0.00318131834134, 0.00131955648461, 0.00046640341759,
    0.01706954162243, 0.01342737701584, 0.00900063997939, 0.00514124713667, 0.00250251364222, 0.00103799989504, 0.00036688592278,
    0.01144205592615, 0.00900063997939, 0.00603330940534, 0.00344628101733, 0.00167748505986, 0.00069579213743, 0.00024593098864,
    0.00653580605408, 0.00514124713667, 0.00344628101733, 0.00196854695367, 0.00095819467066, 0.00039744277546, 0.00014047800980,
    0.00318131834134, 0.00250251364222, 0.00167748505986, 0.00095819467066, 0.00046640341759, 0.00019345616757, 0.00006837798818,
    0.00131955648461, 0.00103799989504, 0.00069579213743, 0.00039744277546, 0.00019345616757, 0.00008024231247, 0.00002836202103)]
SURF_MJ = [-12, -7, -2, 3] * 4                                # createDescriptors_kernel.cl:207-211
SURF_MI = [-12] * 4 + [-7] * 4 + [-2] * 4 + [3] * 4           # createDescriptors_kernel.cl:213-217


def _cl_round(v):
    """OpenCL round() of a float32: half away from zero, result float32."""
    v = float(v)
    return f32(math.floor(v + 0.5) if v >= 0 else -math.floor(-v + 0.5))


def _cdiv(a, b):
    """C integer division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


class _SurfImage:
    def __init__(self, integral):
        self.h, self.w = integral.shape
        self.d = np.ascontiguousarray(integral, np.float32).reshape(-1)

    def box(self, row, col, rows, cols):
        """BoxIntegral, the buffer variant (getOrientation_kernels.cl:73-108)."""
        A = B = Cc = D = f32(0.0)
        r1 = min(row, self.h) - 1
        c1 = min(col, self.w) - 1
        r2 = min(row + rows, self.h) - 1
        c2 = min(col + cols, self.w) - 1
        if r1 >= 0 and c1 >= 0:
            A = self.d[r1 * self.w + c1]
        if r1 >= 0 and c2 >= 0:
            B = self.d[r1 * self.w + c2]
        if r2 >= 0 and c1 >= 0:
            Cc = self.d[r2 * self.w + c1]
        if r2 >= 0 and c2 >= 0:
            D = self.d[r2 * self.w + c2]
        return max(f32(0.0), f32(f32(f32(A - B) - Cc) + D))

    def haar_x(self, row, column, s):
        return f32(self.box(row - _cdiv(s, 2), column, s, _cdiv(s, 2)) -
                   self.box(row - _cdiv(s, 2), column - _cdiv(s, 2), s, _cdiv(s, 2)))

    def haar_y(self, row, column, s):
        return f32(self.box(row, column - _cdiv(s, 2), _cdiv(s, 2), s) -
                   self.box(row - _cdiv(s, 2), column - _cdiv(s, 2), _cdiv(s, 2), s))


def _surf_angle(X, Y):
    """getAngle (getOrientation_kernels.cl:138-153)."""
    L = _m()
    X, Y = f32(X), f32(Y)
    with np.errstate(divide="ignore", invalid="ignore"):
        if X >= 0 and Y >= 0:
            return f32(L.orc_math_atan_pos(f32(Y / X)))
        if X < 0 and Y >= 0:
            return f32(SURF_PI - f32(L.orc_math_atan_pos(f32(f32(-Y) / X))))
        if X < 0 and Y < 0:
            return f32(SURF_PI + f32(L.orc_math_atan_pos(f32(Y / X))))
        if X >= 0 and Y < 0:
            return f32(f32(f32(2) * SURF_PI) - f32(L.orc_math_atan_pos(f32(f32(-Y) / X))))
    return f32(0)


def _surf_gaussian(x, y, sig):
    """gaussian (getOrientation_kernels.cl:68-71): (1/(2 pi sig sig)) * exp(-(x x + y y)/(2 sig sig))."""
    L = _m()
    x, y, sig = f32(x), f32(y), f32(sig)
    norm = f32(f32(1.0) / f32(f32(f32(f32(2.0) * SURF_PI) * sig) * sig))
    arg = f32(f32(-f32(f32(x * x) + f32(y * y))) / f32(f32(f32(2.0) * sig) * sig))
    return f32(norm * f32(L.orc_math_expf(arg)))


def surf_orientation(integral, x, y, scale):
    """getOrientationStep1 + Step2 for one interest point -> float32 orientation."""
    img = _SurfImage(integral)
    s = int(_cl_round(scale))
    r = int(_cl_round(y))
    c = int(_cl_round(x))
    res = []
    for local_id in range(169):
        i = local_id // 13 - 6
        j = local_id % 13 - 6
        if i * i + j * j < 36:
            gauss = SURF_GAUSS25[7 * SURF_ID[i + 6] + SURF_ID[j + 6]]
            rx = f32(gauss * img.haar_x(r + j * s, c + i * s, 4 * s))
            ry = f32(gauss * img.haar_y(r + j * s, c + i * s, 4 * s))
            res.append((rx, ry, _surf_angle(rx, ry)))
    assert len(res) == 109
    sums = []
    two_pi = f32(f32(2) * SURF_PI)
    for tid in range(42):
        ang1 = f32(f32(0.15) * f32(tid))
        third = f32(SURF_PI / f32(3.0))
        ang2 = f32(ang1 - f32(f32(f32(5.0) * SURF_PI) / f32(3.0))) if f32(ang1 + third) > two_pi else f32(ang1 + third)
        sx = sy = f32(0.0)
        for rx, ry, ang in res:
            if tid <= 34:
                check = 1 if (ang1 < ang and ang < ang2) else 0
            else:
                check = 1 if ((ang > 0 and ang < ang2) or (ang > ang1 and ang < two_pi)) else 0
            sx = f32(sx + f32(rx * f32(check)))
            sy = f32(sy + f32(ry * f32(check)))
        sums.append((sx, sy))

    def mag(v):
        return f32(f32(v[0] * v[0]) + f32(v[1] * v[1]))

    offset = 32
    while offset > 0:
        for tid in range(42):                      # all work-items of a step read before the barrier
            if tid < offset and tid + offset < 42 and mag(sums[tid]) < mag(sums[tid + offset]):
                sums[tid] = sums[tid + offset]
        offset >>= 1
    return _surf_angle(sums[0][0], sums[0][1])


def surf_descriptor(integral, x, y, scale, orient):
    """createDescriptors + normalizeDescriptors for one interest point -> 64 float32."""
    L = _m()
    img = _SurfImage(integral)
    px = _cl_round(x)
    py = _cl_round(y)
    orient = f32(orient)
    co = f32(L.orc_math_cosf(orient))
    si = f32(L.orc_math_sinf(orient))
    th = f32(scale)
    desc = np.zeros((16, 4), np.float32)
    lengths = []
    for b in range(16):
        cx = f32(f32(0.5) + f32(b // 4))
        cy = f32(f32(0.5) + f32(b % 4))
        j, i = SURF_MJ[b], SURF_MI[b]
        ix, jx = f32(i + 5), f32(j + 5)
        xs = _cl_round(f32(px + f32(f32(f32(f32(-jx) * th) * si) + f32(f32(ix * th) * co))))
        ys = _cl_round(f32(py + f32(f32(f32(jx * th) * co) + f32(f32(ix * th) * si))))
        d = np.zeros((81, 4), np.float32)
        for tha in range(81):
            k = i + tha // 9
            l = j + tha % 9
            sample_x = int(_cl_round(f32(px + f32(f32(f32(f32(-l) * th) * si) + f32(f32(f32(k) * th) * co)))))
            sample_y = int(_cl_round(f32(py + f32(f32(f32(f32(l) * th) * co) + f32(f32(f32(k) * th) * si)))))
            g1 = _surf_gaussian(f32(xs - f32(sample_x)), f32(ys - f32(sample_y)), f32(f32(2.5) * th))
            hs = int(f32(f32(2) * _cl_round(th)))
            rx = img.haar_x(sample_y, sample_x, hs)
            ry = img.haar_y(sample_y, sample_x, hs)
            rrx = f32(g1 * f32(f32(f32(-rx) * si) + f32(ry * co)))
            rry = f32(g1 * f32(f32(rx * co) + f32(ry * si)))
            d[tha] = (rrx, rry, abs(rrx), abs(rry))
        # sumDesc: items 64..80 fold onto 0..16, then strides 32, 16, ..., 1
        d[:81 - 64] = d[:81 - 64] + d[64:81]
        stride = 32
        while stride > 0:
            d[:stride] = d[:stride] + d[stride:2 * stride]
            stride >>= 1
        g2 = _surf_gaussian(f32(cx - f32(2.0)), f32(cy - f32(2.0)), f32(1.5))
        v = d[0] * g2
        desc[b] = v
        lengths.append(f32(f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2])) + f32(v[3] * v[3])))
    n = 8
    while n > 0:
        for tid in range(n):
            lengths[tid] = f32(lengths[tid] + lengths[tid + n])
        n >>= 1
    inv = f32(f32(1.0) / np.sqrt(lengths[0]))
    return (desc.reshape(64) * inv).astype(np.float32)
