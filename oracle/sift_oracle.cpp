// TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference SIFT path.
//
// This file is the checker the HIP path is compared against. Nothing in the
// product may call it; only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py load the library built from it.
//
// PARITY UNPINNED: the reference holds no golden vectors or known-answer tests
// for SIFT (its tests only cover the OpenCL driver), and its kernels cannot run
// here (no OpenCL device, no OpenCV). This restatement follows the reference
// source line by line instead; every function cites the lines it follows.
//
// Conventions fixed here because the reference leaves them open:
//  * transcendental functions (native_exp/powr/sin/cos, atan2 under
//    -cl-fast-relaxed-math, sift.cpp:1369-1370) come from mif_math.h;
//  * a*b+c written in a tap loop is one fmaf, everything else rounds after each
//    operation in source order (the reference enables -cl-mad-enable, which
//    allows either);
//  * keypoints are listed in raster order (level, y, x) of the detecting work
//    item instead of atomic_inc arrival order (sift.cpp:407), and the list is
//    cut at max_features = items/100 (sift.cpp:1356, 408);
//  * a descriptor cell whose clipped window has negative width AND negative
//    height (sift.cpp:645-649 would then index outside the image) is empty.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "orc_math.h"

namespace {

// sift.cpp:15-39
constexpr int INTVLS = 3;
constexpr int MAX_SCALES = INTVLS + 3;
constexpr int MAX_OCTAVES = 3;
constexpr int IMG_BORDER = 5;
constexpr int MAX_INTERP_STEPS = 5;
constexpr int ORI_BINS = 36;
constexpr int ORI_LANES = 16;
// constants as the kernels see them: text with 6 significant digits, parsed as
// double literals by the OpenCL C compiler (sift.cpp:1372-1394)
constexpr double K_SIGMA = 1.6;
constexpr double K_CONTR_THR_1 = 0.0108333;
constexpr double K_CONTR_THR = 0.065;
constexpr float K_RESAMPLE_RATIO = 2.0f;   // _resample_ratio_x / _y, sift.cpp:812-813
constexpr double K_ORI_SIG_FCTR = 1.5;
constexpr double K_ORI_RADIUS = 4.5;
constexpr double K_ORI_PEAK_RATIO = 0.8;
constexpr double K_DESCR_MAG_THR = 0.2;
constexpr double K_INT_DESCR_FCTR = 512;
constexpr double K_M_PI = 3.14159265358979323846;   // sift.cpp:48
constexpr double K_M_PI2 = 2.0 * K_M_PI;            // sift.cpp:50 (2.0F * M_PI)
constexpr double K_RPI = 4.0 / K_M_PI;              // sift.cpp:51

struct Kp { float x, y, sigma, level; };

struct Octave {
    int w = 0, h = 0, id = 0;
    uint32_t max_feat = 0;
    uint32_t raw_count = 0;           // what the atomic counter would hold
    std::vector<float> pyr;           // [6][h][w]
    std::vector<float> scratch;       // [h][w]
    std::vector<float> dog;           // [5][h][w]
    std::vector<Kp> kp;
    std::vector<float> orient;
    std::vector<float> desc;
};

struct Sift {
    int W = 0, H = 0, C = 1;
    bool up = false;                     // first octave at twice the input size (sift.cpp:1306-1310)
    std::vector<float> taps[MAX_SCALES];
    int fr[MAX_SCALES];
    Octave oct[MAX_OCTAVES];
};

// sift.cpp:819-889 Build_separable_filter_cof
void build_taps(Sift& S)
{
    const float sigma = 1.6f;
    float sig[MAX_SCALES];
    sig[0] = sigma;
    const float k = powf(2.0f, 1.0f / (INTVLS));
    for (uint32_t s = 1; s < MAX_SCALES; s++) {
        float prev = sigma * powf(k, (float)(s - 1));
        float next = prev * k;
        sig[s] = (float)std::sqrt(next * next - prev * prev);
    }
    for (uint32_t s = 0; s < MAX_SCALES; s++) {
        float sum = 0.0f, fcval = 0.0f;
        const size_t frad = (size_t)(std::ceil(3.5f * sig[s]));
        std::vector<float>& f = S.taps[s];
        f.assign(2 * frad + 1, 0.0f);
        size_t fc = 0;
        for (fc = 0; fc < frad; fc++) {
            f[fc] = f[2 * frad - fc] = fcval =
                float(std::exp(float(-0.5 * ((fc - frad) * (fc - frad) / (sig[s] * sig[s])))));
            sum += 2 * fcval;
        }
        f[fc] = 1;
        sum += 1;
        fcval = 1.0f / sum;
        for (fc = 0; fc < 2 * frad + 1; fc++) f[fc] *= fcval;
        S.fr[s] = (int)frad;          // kernels get width/2 (sift.cpp:1198)
    }
}

// sift.cpp:229-243 kBlurH : pyramid level `ssid` -> scratch.
// Per pixel: prod = 0; for i = -fr .. fr-1: prod = fma(filter[fr+i], in[y][x+i] or 0, prod).
// The loops are ordered tap-outer / x-inner so gcc vectorises them; each pixel
// still accumulates its taps in ascending order, and taps that fall outside
// the image add an exact zero, so skipping them changes no bit.
void blur_h(const float* in, const float* filter, float* out, int w, int h, int fr)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        const float* row = in + (size_t)y * w;
        float* o = out + (size_t)y * w;
        for (int x = 0; x < w; x++) o[x] = 0.0f;
        for (int i = -fr; i < fr; i++) {
            const float f = filter[fr + i];
            const int x0 = std::max(0, -i), x1 = std::min(w, w - i);
            for (int x = x0; x < x1; x++) o[x] = fmaf(f, row[x + i], o[x]);
        }
    }
}

// sift.cpp:213-227 kBlurV : scratch -> pyramid level `ssid`
void blur_v(const float* in, const float* filter, float* out, int w, int h, int fr)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        float* o = out + (size_t)y * w;
        for (int x = 0; x < w; x++) o[x] = 0.0f;
        for (int i = -fr; i < fr; i++) {
            if (y + i < 0 || y + i > h - 1) continue;
            const float f = filter[fr + i];
            const float* row = in + (size_t)(y + i) * w;
            for (int x = 0; x < w; x++) o[x] = fmaf(f, row[x], o[x]);
        }
    }
}

inline float at(const float* p, int w, int y, int x) { return p[y * w + x]; }

// sift.cpp:292-322 LOC_EXTREMUM, including the repeated (y-1,x-1) of the
// previous image and the missing (y-1,x+1).
template <bool GT>
bool loc_extremum(float v, const float* p, const float* c, const float* n, int w, int y, int x)
{
    auto cmp = [&](float o) { return GT ? (v > o) : (v < o); };
    return cmp(at(p, w, y, x - 1)) && cmp(at(p, w, y, x)) && cmp(at(p, w, y, x + 1)) &&
           cmp(at(p, w, y + 1, x - 1)) && cmp(at(p, w, y + 1, x)) && cmp(at(p, w, y + 1, x + 1)) &&
           cmp(at(p, w, y - 1, x - 1)) && cmp(at(p, w, y - 1, x)) && cmp(at(p, w, y - 1, x - 1)) &&
           cmp(at(c, w, y, x - 1)) && cmp(at(c, w, y, x + 1)) &&
           cmp(at(c, w, y + 1, x - 1)) && cmp(at(c, w, y + 1, x)) && cmp(at(c, w, y + 1, x + 1)) &&
           cmp(at(c, w, y - 1, x - 1)) && cmp(at(c, w, y - 1, x)) && cmp(at(c, w, y - 1, x + 1)) &&
           cmp(at(n, w, y, x - 1)) && cmp(at(n, w, y, x)) && cmp(at(n, w, y, x + 1)) &&
           cmp(at(n, w, y + 1, x - 1)) && cmp(at(n, w, y + 1, x)) && cmp(at(n, w, y + 1, x + 1)) &&
           cmp(at(n, w, y - 1, x - 1)) && cmp(at(n, w, y - 1, x)) && cmp(at(n, w, y - 1, x + 1));
}

// sift.cpp:265-417 kDetector for one work item; returns true and fills `out`
// when the item would append a feature.
bool detect_one(const float* dog_p, const float* dog_c, const float* dog_n, int w, int h, int s, int o,
                int x, int y, Kp* out)
{
    if (y <= IMG_BORDER) return false;
    if (y >= h - IMG_BORDER) return false;
    if (x <= IMG_BORDER) return false;
    if (x >= w - IMG_BORDER) return false;

    float curr_pix = dog_c[y * w + x];
    if ((double)fabsf(curr_pix) <= K_CONTR_THR_1) return false;

    if (!(loc_extremum<true>(curr_pix, dog_p, dog_c, dog_n, w, y, x) ||
          loc_extremum<false>(curr_pix, dog_p, dog_c, dog_n, w, y, x)))
        return false;

    float Dxx = 0, Dyy = 0, Dss, Dxy = 0, Dxs, Dys, trH, detH;
    float gx = 0, gy = 0, gz = 0;
    float ox = 0.0f, oy = 0.0f, oz = 0.0f;
    int ty = y, tx = x, ts = s, iter;

    for (iter = 0; iter < MAX_INTERP_STEPS; iter++) {
        gx = 0.5f * (dog_c[ty * w + tx + 1] - dog_c[ty * w + tx - 1]);
        gy = 0.5f * (dog_c[(ty + 1) * w + tx] - dog_c[(ty - 1) * w + tx]);
        gz = 0.5f * (dog_n[ty * w + tx] - dog_p[ty * w + tx]);

        Dxx = dog_c[ty * w + tx + 1] - 2.0f * dog_c[ty * w + tx] + dog_c[ty * w + tx - 1];
        Dyy = dog_c[(ty + 1) * w + tx] - 2.0f * dog_c[ty * w + tx] + dog_c[(ty - 1) * w + tx];
        Dss = dog_n[ty * w + tx] - 2.0f * dog_c[ty * w + tx] + dog_p[ty * w + tx];

        Dxy = 0.25f * (dog_c[(ty + 1) * w + tx + 1] + dog_c[(ty - 1) * w + tx - 1] -
                       dog_c[(ty + 1) * w + tx - 1] - dog_c[(ty - 1) * w + tx + 1]);
        Dxs = 0.25f * (dog_n[ty * w + tx + 1] + dog_p[ty * w + tx - 1] - dog_n[ty * w + tx - 1] -
                       dog_p[ty * w + tx + 1]);
        Dys = 0.25f * (dog_n[(ty + 1) * w + tx] + dog_p[(ty - 1) * w + tx] - dog_p[(ty + 1) * w + tx] -
                       dog_n[(ty - 1) * w + tx]);

        float h0x = Dyy * Dss - Dys * Dys;
        float h1x = Dys * Dxs - Dxy * Dss;
        float h2x = Dxy * Dys - Dyy * Dxs;
        detH = Dxx * h0x + Dxy * h1x + Dxs * h2x;

        float h0y = Dxs * Dys - Dxy * Dss;
        float h0z = Dxy * Dys - Dxs * Dyy;
        float h1y = Dxx * Dss - Dxs * Dxs;
        float h1z = Dxs * Dxy - Dxx * Dys;
        float h2y = Dxy * Dxs - Dxx * Dys;
        float h2z = Dxx * Dyy - Dxy * Dxy;

        if (detH == 0.0f) return false;
        const float dv = -1.0f / detH;

        h0x *= dv; h0y *= dv; h0z *= dv;
        h1x *= dv; h1y *= dv; h1z *= dv;
        h2x *= dv; h2y *= dv; h2z *= dv;

        ox = h0x * gx + h0y * gy + h0z * gz;
        oy = h1x * gx + h1y * gy + h1z * gz;
        oz = h2x * gx + h2y * gy + h2z * gz;

        if (fabsf(ox) < 0.5f && fabsf(oy) < 0.5f && fabsf(oz) < 0.5f) break;

        tx += (int)mif_roundf(ox);
        if (tx < IMG_BORDER || tx >= w - IMG_BORDER) return false;
        ty += (int)mif_roundf(oy);
        if (ty < IMG_BORDER || ty >= h - IMG_BORDER) return false;
        ts += (int)mif_roundf(oz);
        if (ts < 1 || ts > INTVLS) return false;
    }
    if (iter >= MAX_INTERP_STEPS) return false;

    trH = Dxx + Dyy;
    detH = Dxx * Dyy - Dxy * Dxy;
    if ((double)fabsf(detH) <= 1e-8) return false;
    // ((10+1)*(10+1)/10) is integer arithmetic: 12
    if ((trH * trH) / detH >= (float)((10 + 1) * (10 + 1) / 10)) return false;

    curr_pix = ts == 1 ? dog_p[ty * w + tx] : ts == 2 ? dog_c[ty * w + tx] : dog_n[ty * w + tx];
    detH = curr_pix + 0.5f * (gx * ox + gy * oy + gz * oz);

    if (fabsf(detH) > (float)(K_CONTR_THR / INTVLS)) {
        const float p2o = mif_pow2i(o);
        out->x = ((float)tx + ox) * p2o;
        out->y = ((float)ty + oy) * p2o;
        out->sigma = (float)(K_SIGMA * (double)mif_exp2f(((float)ts + oz) / (float)INTVLS));
        out->level = (float)ts;
        return true;
    }
    return false;
}

// sift.cpp:431-582 kOrientation for one work-group (one keypoint, 16 lanes)
float orientation_one(const Kp& f, const float* pyr, int w, int h, int oid)
{
    float hist_sub[ORI_LANES][ORI_BINS];
    for (int l = 0; l < ORI_LANES; l++)
        for (int i = 0; i < ORI_BINS; i++) hist_sub[l][i] = 0.0f;

    const float exp_denom = (float)((double)f.sigma * K_ORI_SIG_FCTR);
    const int rad = (int)mif_round(K_ORI_RADIUS * (double)f.sigma);
    const float exp_factor = -0.5f / (exp_denom * exp_denom);
    const int sq_thres = rad * rad;
    const int offset_py = (int)((float)(w * h) * f.level);
    const float p2o = mif_pow2i(oid);
    const int x = (int)(f.x / p2o);
    const int y = (int)(f.y / p2o);

    const int xmin = std::max(1, x - rad);
    const int ymin = std::max(1, y - rad);
    const int xmax = std::min(w - 2, x + rad);
    const int ymax = std::min(h - 2, y + rad);
    const int wx = xmax - xmin + 1;
    const int hy = ymax - ymin + 1;
    const int loops = wx * hy;

    for (int lid = 0; lid < ORI_LANES; lid++) {
        for (int i = lid; i < loops; i += ORI_LANES) {
            const int yy = i / wx + ymin;
            const int xx = i % wx + xmin;
            if (xx > 0 && xx < w - 1 && yy > 0 && yy < h - 1) {
                float dx = pyr[yy * w + xx + 1 + offset_py] - pyr[yy * w + xx - 1 + offset_py];
                float dy = pyr[(yy - 1) * w + xx + offset_py] - pyr[(yy + 1) * w + xx + offset_py];
                const float grad = sqrtf(dx * dx + dy * dy);
                const float theta = mif_atan2f(dy, dx);
                dx = (float)(xx - x);
                dy = (float)(yy - y);
                const int sq_dist = (int)(dx * dx + dy * dy);
                if (sq_dist > sq_thres) continue;
                int bin = (int)mif_round((ORI_BINS * ((double)theta + K_M_PI)) / K_M_PI2);
                bin = (bin < ORI_BINS) ? bin : 0;
                hist_sub[lid][bin] += grad * mif_expf((float)sq_dist * exp_factor);
            }
        }
    }
    // tree reduction, sift.cpp:509-522
    for (int i = 0; i < ORI_BINS; i++) {
        for (int l = 0; l < 8; l++) hist_sub[l][i] += hist_sub[l + 8][i];
        for (int l = 0; l < 4; l++) hist_sub[l][i] += hist_sub[l + 4][i];
        for (int l = 0; l < 2; l++) hist_sub[l][i] += hist_sub[l + 2][i];
        hist_sub[0][i] += hist_sub[1][i];
    }
    float* hs = hist_sub[0];
    float hp, hc, h0, hn, fbin, v;
    // smoothing in double because of the 0.25 / 0.5 literals, sift.cpp:531-543
    for (int j = 0; j < 2; j++) {
        h0 = hs[0];
        hp = hs[ORI_BINS - 1];
        for (int i = 0; i < ORI_BINS; i++) {
            hc = hs[i];
            hs[i] = (float)(0.25 * (double)hp + 0.5 * (double)hs[i] +
                            0.25 * (double)((i + 1 == ORI_BINS) ? h0 : hs[i + 1]));
            hp = hc;
        }
    }
    v = hs[0];
    for (int i = 1; i < ORI_BINS; i++)
        if (hs[i] > v) v = hs[i];
    float result = -1.0f;
    v = (float)(K_ORI_PEAK_RATIO * (double)v);
    for (int i = 0; i < ORI_BINS; i++) {
        hc = hs[i];
        hp = hs[(i == 0) ? (ORI_BINS - 1) : (i - 1)];
        hn = hs[(i + 1) % ORI_BINS];
        if (hc >= v && hc > hn && hc > hp) {
            fbin = (float)((double)(float)i +
                           ((double)(0.5f * (hp - hn)) / ((double)hp - 2.0 * (double)hc + (double)hn)));
            fbin = ((double)fbin < 0.0) ? (fbin + (float)ORI_BINS)
                                        : (((double)fbin >= (double)(float)ORI_BINS) ? (fbin - (float)ORI_BINS) : fbin);
            result = (float)(((K_M_PI2 * (double)fbin) / ORI_BINS) - K_M_PI);
        }
    }
    return result;
}

// sift.cpp:585-709 kDescriptor for one work-group (one keypoint, 16 cells x 8 lanes)
void descriptor_one(const Kp& f, float orientation, const float* pyr, int w, int h, int oid, float* out128)
{
    float des[16][8][9];
    memset(des, 0, sizeof(des));
    const float p2o = mif_pow2i(oid);
    const int x = (int)(f.x / p2o);
    const int y = (int)(f.y / p2o);
    const int offset_py = (int)((float)(w * h) * f.level);
    const float ang = -orientation;
    const float spt = (float)fabs((double)f.sigma * 3.0);
    float s, c;
    mif_sincosf(ang, &s, &c);
    const float cspt = c * spt;
    const float sspt = s * spt;
    const float crspt = c / spt;
    const float srspt = s / spt;

    for (int bid = 0; bid < 16; bid++) {
        const int ix = bid & 0x3;
        const int iy = bid >> 2;
        const float offsetptx = (float)ix - 1.5f;
        const float offsetpty = (float)iy - 1.5f;
        const float ptx = cspt * offsetptx - sspt * offsetpty + (float)x;
        const float pty = cspt * offsetpty + sspt * offsetptx + (float)y;
        const float bsz = fabsf(cspt) + fabsf(sspt);
        // MAXV/MINV compare int with float in float and convert back by truncation
        const float fxmin = floorf(ptx - bsz), fymin = floorf(pty - bsz);
        const float fxmax = floorf(ptx + bsz), fymax = floorf(pty + bsz);
        const int xmin = (1.0f >= fxmin) ? 1 : (int)fxmin;
        const int ymin = (1.0f >= fymin) ? 1 : (int)fymin;
        const int xmax = ((float)(w - 2) <= fxmax) ? (w - 2) : (int)fxmax;
        const int ymax = ((float)(h - 2) <= fymax) ? (h - 2) : (int)fymax;
        const int wx = xmax - xmin + 1;
        const int hy = ymax - ymin + 1;
        const int loops = (wx > 0 && hy > 0) ? wx * hy : 0;
        for (int lid = 0; lid < 8; lid++) {
            float* d = des[bid][lid];
            for (int i = lid; i < loops; i += 8) {
                const int yy = i / wx + ymin;
                const int xx = i % wx + xmin;
                float dx = (float)xx - ptx;
                float dy = (float)yy - pty;
                const float nx = crspt * dx + srspt * dy;
                const float ny = crspt * dy - srspt * dx;
                const float nxn = fabsf(nx);
                const float nyn = fabsf(ny);
                if (nxn < 1.0f && nyn < 1.0f) {
                    dx = pyr[yy * w + xx + 1 + offset_py] - pyr[yy * w + xx - 1 + offset_py];
                    dy = pyr[(yy + 1) * w + xx + offset_py] - pyr[(yy - 1) * w + xx + offset_py];
                    const float mod = sqrtf(dx * dx + dy * dy);
                    float th = mif_atan2f(dy, dx);
                    const float dnx = nx + offsetptx;
                    const float dny = ny + offsetpty;
                    const float ww = mif_expf(-0.125f * (dnx * dnx + dny * dny));
                    const float wwx = (float)(1.0 - (double)nxn);
                    const float wwy = (float)(1.0 - (double)nyn);
                    const float wgt = ww * wwx * wwy * mod;
                    th -= ang;
                    while (th < 0.0f) th = (float)((double)th + K_M_PI2);
                    while ((double)th >= K_M_PI2) th = (float)((double)th - K_M_PI2);
                    const float tth = (float)((double)th * K_RPI);
                    const int fo0 = (int)floorf(tth);
                    const float do0 = tth - (float)fo0;
                    const float wgt1 = 1.0f - do0;
                    const float wgt2 = do0;
                    const int fo = fo0 % 8;
                    if (fo < 8) {
                        d[fo] += (wgt1 * wgt);
                        d[fo + 1] += (wgt2 * wgt);
                    }
                }
            }
            d[0] += d[8];
        }
    }
    for (int bid = 0; bid < 16; bid++)
        for (int lid = 0; lid < 8; lid++) {
            float bin = 0;
            for (int i = 0; i < 8; i++) bin += des[bid][i][lid];
            out128[bid * 8 + lid] = bin;
        }
}

inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// sift.cpp:712-765 kNormalize for one work item
void normalize_one(float* d)
{
    float tmp[128];
    float norm1 = 0.0f, norm2 = 0.0f;
    for (int i = 0; i < 128; i++) tmp[i] = d[i] * d[i];
    for (int i = 0; i < 128; i++) norm1 += tmp[i];
    norm1 = 1.0f / sqrtf(norm1);
    for (int i = 0; i < 128; i++) {
        float v1 = d[i] * norm1;
        float v0 = clampf(v1, 0.0f, (float)K_DESCR_MAG_THR);
        d[i] = v0;
        tmp[i] = v0 * v0;
    }
    for (int i = 0; i < 128; i++) norm2 += tmp[i];
    norm2 = 1.0f / sqrtf(norm2);
    for (int i = 0; i < 128; i++) {
        float v1 = d[i] * norm2;
        float v0 = (float)K_INT_DESCR_FCTR * v1;
        d[i] = clampf(v0, 0.0f, 255.0f);
    }
}

// sift.cpp:1119-1244 call_async_detector for one octave (after the base image
// is in pyramid level 0)
void run_octave_detector(Sift& S, Octave& O)
{
    const int w = O.w, h = O.h;
    const size_t items = (size_t)w * h;
    for (int s = 0; s < MAX_SCALES; s++) {
        const int src = s < 1 ? 0 : s - 1;
        blur_h(&O.pyr[src * items], S.taps[s].data(), O.scratch.data(), w, h, S.fr[s]);
        blur_v(O.scratch.data(), S.taps[s].data(), &O.pyr[s * items], w, h, S.fr[s]);
    }
    // sift.cpp:248-261 kDiff
    for (int i = 0; i < MAX_SCALES - 1; i++) {
        const float* prev = &O.pyr[i * items];
        const float* next = &O.pyr[(i + 1) * items];
        float* out = &O.dog[i * items];
#pragma omp parallel for schedule(static)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) out[y * w + x] = next[y * w + x] - prev[y * w + x];
    }
    O.kp.clear();
    O.raw_count = 0;
    std::vector<std::vector<Kp>> rows((size_t)INTVLS * h);
    for (int s = 1; s <= INTVLS; s++) {
        const float* dp = &O.dog[(s - 1) * items];
        const float* dc = &O.dog[s * items];
        const float* dn = &O.dog[(s + 1) * items];
#pragma omp parallel for schedule(dynamic, 8)
        for (int y = 0; y < h; y++) {
            std::vector<Kp>& r = rows[(size_t)(s - 1) * h + y];
            for (int x = 0; x < w; x++) {
                Kp k;
                if (detect_one(dp, dc, dn, w, h, s, O.id, x, y, &k)) r.push_back(k);
            }
        }
    }
    for (auto& r : rows)
        for (auto& k : r) {
            if (O.raw_count < O.max_feat) O.kp.push_back(k);
            O.raw_count++;
        }
}

void run_octave_descriptor(Octave& O)
{
    const int n = (int)O.kp.size();
    O.orient.assign(n, 0.0f);
    O.desc.assign((size_t)n * 128, 0.0f);
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < n; i++) {
        O.orient[i] = orientation_one(O.kp[i], O.pyr.data(), O.w, O.h, O.id);
        descriptor_one(O.kp[i], O.orient[i], O.pyr.data(), O.w, O.h, O.id, &O.desc[(size_t)i * 128]);
        normalize_one(&O.desc[(size_t)i * 128]);
    }
}

} // namespace

extern "C" {

void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_get_threads()
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// sift.cpp:1291-1366 sift_octave ctor (sizes) and 1731-1770 sift_algorithm ctor
void* orc_sift_create_up(int width, int height, int channels, int up_scale);

void* orc_sift_create(int width, int height, int channels) { return orc_sift_create_up(width, height, channels, 0); }

void* orc_sift_create_up(int width, int height, int channels, int up_scale)
{
    if (width <= 0 || height <= 0 || (channels != 1 && channels != 3)) return nullptr;
    Sift* S = new Sift();
    S->W = width; S->H = height; S->C = channels; S->up = up_scale != 0;
    build_taps(*S);
    for (int o = 0; o < MAX_OCTAVES; o++) {
        Octave& O = S->oct[o];
        O.id = o;
        if (S->up) {
            // sift.cpp:1306-1310: _resample_ratio (2.0f) * size_t / std::pow(2, id), truncated
            O.h = (int)(size_t)(K_RESAMPLE_RATIO * (size_t)height / std::pow(2, o));
            O.w = (int)(size_t)(K_RESAMPLE_RATIO * (size_t)width / std::pow(2, o));
        } else {
            O.h = (int)(size_t)(height / std::pow(2, o));
            O.w = (int)(size_t)(width / std::pow(2, o));
        }
        const size_t items = (size_t)O.w * O.h;
        O.max_feat = (uint32_t)(items / 100);
        O.pyr.assign(items * MAX_SCALES, 0.0f);
        O.scratch.assign(items, 0.0f);
        O.dog.assign(items * (MAX_SCALES - 1), 0.0f);
    }
    return S;
}

void orc_sift_destroy(void* hdl) { delete (Sift*)hdl; }

// sift.cpp:1856-1916 sift_algorithm::extract_features. Returns 0, or -1 when an
// octave yields no keypoint (sift.cpp:1257-1258); later octaves are then not run.
int orc_sift_run(void* hdl, const uint8_t* pixels, int want_descriptors)
{
    Sift& S = *(Sift*)hdl;
    for (int o = 0; o < MAX_OCTAVES; o++) { S.oct[o].kp.clear(); S.oct[o].orient.clear(); S.oct[o].desc.clear(); S.oct[o].raw_count = 0; }
    for (int o = 0; o < MAX_OCTAVES; o++) {
        Octave& O = S.oct[o];
        const int w = O.w, h = O.h;
        if (o == 0 && S.up) {
            // sift.cpp:1140-1177: kResize_NN_3ch_8b then kColor_interleaved_ch, or
            // copy_pixels_1ch8_to_1ch32 (into the non-resampled buffer, sift.cpp:907) then
            // kResize_NN_1ch_32b (sift.cpp:134-210); source size as sift.cpp:1144-1145
            float* dst = O.pyr.data();
            const int ws = (int)((float)w / K_RESAMPLE_RATIO), hs = (int)((float)h / K_RESAMPLE_RATIO);
            const float cb = (float)0.114, cg = (float)0.587, cr = (float)0.299;
            const float val_scale = (float)0.0039215686;
            for (int y = 0; y < h; y++) {
                for (int x = 0; x < w; x++) {
                    const float fy = ((float)y + 0.4995f) / K_RESAMPLE_RATIO;
                    const float fx = ((float)x + 0.4995f) / K_RESAMPLE_RATIO;
                    const int ps = (int)fy * ws + (int)fx;
                    const bool inside = fx < (float)ws && fy < (float)hs;
                    float v;
                    if (S.C == 3) {
                        const float b = inside ? (float)pixels[3 * ps] : 0.0f;
                        const float g = inside ? (float)pixels[3 * ps + 1] : 0.0f;
                        const float r = inside ? (float)pixels[3 * ps + 2] : 0.0f;
                        v = (b * cb + g * cg + r * cr) * val_scale;
                    } else {
                        v = inside ? (float)pixels[ps] / 255.0f : 0.0f;
                    }
                    dst[y * w + x] = v;
                }
            }
        } else if (o == 0) {
            float* dst = O.pyr.data();
            if (S.C == 3) {
                // sift.cpp:55-72 kColor_interleaved_ch
                const float cb = (float)0.114, cg = (float)0.587, cr = (float)0.299;
                const float val_scale = (float)0.0039215686;
                for (int i = 0; i < w * h; i++) {
                    const float b = (float)pixels[3 * i], g = (float)pixels[3 * i + 1], r = (float)pixels[3 * i + 2];
                    dst[i] = (b * cb + g * cg + r * cr) * val_scale;
                }
            } else {
                // sift.cpp:901-913 copy_pixels_1ch8_to_1ch32
                for (int i = 0; i < w * h; i++) dst[i] = (float)pixels[i] / 255.0f;
            }
        } else {
            // sift.cpp:113-132 kDown_pyramid with wo = 2w, d = 2, ss_img_id = 3 (sift.cpp:1187-1191)
            const float* src = S.oct[o - 1].pyr.data();
            const int wo = w * 2, d = 2;
            float* dst = O.pyr.data();
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    const int offset_ss_image = INTVLS * w * d * h * d;
                    dst[y * w + x] = src[(y * d) * wo + x * d + offset_ss_image];
                }
        }
        run_octave_detector(S, O);
        if (want_descriptors) {
            if (O.raw_count < 1) return -1;
            run_octave_descriptor(O);
        }
    }
    return 0;
}

int orc_sift_octave_width(void* hdl, int o) { return ((Sift*)hdl)->oct[o].w; }
int orc_sift_octave_height(void* hdl, int o) { return ((Sift*)hdl)->oct[o].h; }
int orc_sift_max_features(void* hdl, int o) { return (int)((Sift*)hdl)->oct[o].max_feat; }
int orc_sift_count(void* hdl, int o) { return (int)((Sift*)hdl)->oct[o].kp.size(); }
int orc_sift_raw_count(void* hdl, int o) { return (int)((Sift*)hdl)->oct[o].raw_count; }
const float* orc_sift_pyramid(void* hdl, int o) { return ((Sift*)hdl)->oct[o].pyr.data(); }
const float* orc_sift_dog(void* hdl, int o) { return ((Sift*)hdl)->oct[o].dog.data(); }
const float* orc_sift_keypoints(void* hdl, int o) { return (const float*)((Sift*)hdl)->oct[o].kp.data(); }
const float* orc_sift_orientations(void* hdl, int o) { return ((Sift*)hdl)->oct[o].orient.data(); }
const float* orc_sift_descriptors(void* hdl, int o) { return ((Sift*)hdl)->oct[o].desc.data(); }

int orc_sift_taps(void* hdl, int level, float* out, int cap)
{
    Sift& S = *(Sift*)hdl;
    const int n = (int)S.taps[level].size();
    for (int i = 0; i < n && i < cap; i++) out[i] = S.taps[level][i];
    return S.fr[level];
}

// element-wise access to mif_math.h for the accuracy tests
float orc_math_expf(float x) { return mif_expf(x); }
float orc_math_exp2f(float x) { return mif_exp2f(x); }
float orc_math_atan2f(float y, float x) { return mif_atan2f(y, x); }
float orc_math_sinf(float x) { float s, c; mif_sincosf(x, &s, &c); return s; }
float orc_math_cosf(float x) { float s, c; mif_sincosf(x, &s, &c); return c; }
// the same functions over arrays; op: 0 exp, 1 exp2, 2 sin, 3 cos, 4 atan2(a, b), 5 exp without range guards
void orc_math_array(int op, const float* a, const float* b, float* out, int n)
{
    for (int i = 0; i < n; i++) {
        float s, c, r = 0.0f;
        switch (op) {
        case 0: r = mif_expf(a[i]); break;
        case 1: r = mif_exp2f(a[i]); break;
        case 2: mif_sincosf(a[i], &s, &c); r = s; break;
        case 3: mif_sincosf(a[i], &s, &c); r = c; break;
        case 4: r = mif_atan2f(a[i], b[i]); break;
        case 5: r = mif_expf_bounded(a[i]); break;
        }
        out[i] = r;
    }
}


// SIFT::test_kResize (SIFT/sift.cpp:2097-2117): kResize_NN_1ch_8b / kResize_NN_3ch_8b (sift.cpp:159-206)
// over the (int)(width * ratio_x) x (int)(height * ratio_y) destination.
int orc_resize_u8(const uint8_t* src, int width, int height, int channels, float ratio_x, float ratio_y, uint8_t* dst)
{
    const int wd = (int)((float)width * ratio_x), hd = (int)((float)height * ratio_y);
    for (int y = 0; y < hd; y++)
        for (int x = 0; x < wd; x++) {
            const float fy = ((float)y + 0.4995f) / ratio_y;
            const float fx = ((float)x + 0.4995f) / ratio_x;
            const int pd = y * wd + x;
            const int ps = (int)fy * width + (int)fx;
            const bool inside = fx < (float)width && fy < (float)height;
            for (int c = 0; c < channels; c++) dst[(size_t)pd * channels + c] = inside ? src[(size_t)ps * channels + c] : 0;
        }
    return wd * hd;
}

// SIFT::test_ColorToGray_Resize (SIFT/sift.cpp:2119-2144): kColor_interleaved_ch (sift.cpp:55-72), then
// kResize_NN_1ch_32b (sift.cpp:136-157).
int orc_gray_resize(const uint8_t* bgr, int width, int height, float ratio_x, float ratio_y, float* dst)
{
    std::vector<float> gray((size_t)width * height);
    const float cb = (float)0.114, cg = (float)0.587, cr = (float)0.299;
    const float val_scale = (float)0.0039215686;
    for (size_t i = 0; i < gray.size(); i++) {
        const float b = (float)bgr[3 * i], g = (float)bgr[3 * i + 1], r = (float)bgr[3 * i + 2];
        gray[i] = (b * cb + g * cg + r * cr) * val_scale;
    }
    const int wd = (int)((float)width * ratio_x), hd = (int)((float)height * ratio_y);
    for (int y = 0; y < hd; y++)
        for (int x = 0; x < wd; x++) {
            const float fy = ((float)y + 0.4995f) / ratio_y;
            const float fx = ((float)x + 0.4995f) / ratio_x;
            const int ps = (int)fy * width + (int)fx;
            dst[y * wd + x] = (fx < (float)width && fy < (float)height) ? gray[ps] : 0.0f;
        }
    return wd * hd;
}

} // extern "C"
