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
