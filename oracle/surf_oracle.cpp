// TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference SURF path.
//
// This file is the checker the HIP path is compared against. Nothing in the
// product may call it; only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py load the library built from it.
//
// PARITY UNPINNED: the reference holds no golden vectors or tests for SURF and
// its kernels cannot run here (no OpenCL device, no OpenCV). The restatement
// follows SURF/coopcl_surf/ccl_surf/ccl_surf.cpp and the kernels in
// SURF/coopcl_surf/CLSource/*.cl line by line; every function cites its lines.
//
// Conventions fixed here because the reference leaves them open:
//  * exp / sin / cos / atan come from mif_math.h (the kernels use the OpenCL
//    built-ins, whose bits no test pins);
//  * interest points are listed in the order (non_max_supression launch, row,
//    column) instead of atom_add arrival order (nonMaxSuppression_kernel.cl:357),
//    and the list is cut at 8000 (INIT_FEATS, surf_constants.h);
//  * the 109 orientation samples of a point are stored in ascending work-item
//    order instead of local atom_add order (getOrientation_kernels.cl:224);
//  * hessian_det work items beyond layerWidth / layerHeight do not write (in the
//    reference they write one element past the row, hessianDet_kernel.cl:137,172);
//  * gray conversion of 8-bit input: OpenCV 4.x (un-vendored) cv::cvtColor
//    BGR2GRAY fixed-point formula (b*1868 + g*9617 + r*4899 + 8192) >> 14 and
//    convertTo(CV_32F, 1/255) = float(double(v) * (1.0/255.0))
//    (ccl_surf.cpp:422-449).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cfloat>
#include <vector>
#include <algorithm>
#include "orc_math.h"

namespace {

// surf_constants.h
constexpr int OCTAVES = 3;
constexpr int SAMPLE_STEP = 5;
constexpr float THRES = 0.0001f;
constexpr int INIT_FEATS = 8000;
constexpr float PI_F = 3.14159274101257f;   // M_PI_F

// ccl_surf.cpp:743-762
const int TAB_J[16] = {-12, -7, -2, 3, -12, -7, -2, 3, -12, -7, -2, 3, -12, -7, -2, 3};
const int TAB_I[16] = {-12, -12, -12, -12, -7, -7, -7, -7, -2, -2, -2, -2, 3, 3, 3, 3};
const unsigned TAB_ID[13] = {6, 5, 4, 3, 2, 1, 0, 1, 2, 3, 4, 5, 6};
const float GAUSS25[49] = {
    0.02350693969273f, 0.01849121369071f, 0.01239503121241f, 0.00708015417522f, 0.00344628101733f, 0.00142945847484f, 0.00050524879060f,
    0.02169964028389f, 0.01706954162243f, 0.01144205592615f, 0.00653580605408f, 0.00318131834134f, 0.00131955648461f, 0.00046640341759f,
    0.01706954162243f, 0.01342737701584f, 0.00900063997939f, 0.00514124713667f, 0.00250251364222f, 0.00103799989504f, 0.00036688592278f,
    0.01144205592615f, 0.00900063997939f, 0.00603330940534f, 0.00344628101733f, 0.00167748505986f, 0.00069579213743f, 0.00024593098864f,
    0.00653580605408f, 0.00514124713667f, 0.00344628101733f, 0.00196854695367f, 0.00095819467066f, 0.00039744277546f, 0.00014047800980f,
    0.00318131834134f, 0.00250251364222f, 0.00167748505986f, 0.00095819467066f, 0.00046640341759f, 0.00019345616757f, 0.00006837798818f,
    0.00131955648461f, 0.00103799989504f, 0.00069579213743f, 0.00039744277546f, 0.00019345616757f, 0.00008024231247f, 0.00002836202103f};

struct Layer {
    int w, h, step, filter;
    std::vector<float> resp;
    std::vector<int> lap;
};

struct Ipt {
    float x, y, scale, orientation;
    int laplacian;
    float desc[64];
};

struct Surf {
    int W = 0, H = 0;
    std::vector<float> gray, tmp, integral;
    std::vector<Layer> layers;
    std::vector<Ipt> ipts;
    int raw_count = 0;
};

// integralImage_kernels.cl:63-152 `scan` for one row: Kogge-Stone sub-scans of
// 64 elements with zero padding, plus the running total of the previous blocks
void scan_row(const float* in, float* out, int cols)
{
    constexpr int WG = 64, HALF = 32;
    float lData[WG + HALF], lData2[WG + HALF];
    for (int i = 0; i < HALF; i++) lData[i] = lData2[i] = 0.0f;
    float prevMaxVal = 0;
    int iterations = cols / WG;
    if (cols % WG != 0) iterations++;
    for (int it = 0; it < iterations; it++) {
        for (int t = 0; t < WG; t++) {
            const int col = it * WG + t;
            lData[t + HALF] = col < cols ? in[col] : 0.0f;
        }
        for (int t = 0; t < WG; t++) lData2[t + HALF] = lData[t + HALF] + lData[t + HALF - 1];
        for (int t = 0; t < WG; t++) lData[t + HALF] = lData2[t + HALF] + lData2[t + HALF - 2];
        for (int t = 0; t < WG; t++) lData2[t + HALF] = lData[t + HALF] + lData[t + HALF - 4];
        for (int t = 0; t < WG; t++) lData[t + HALF] = lData2[t + HALF] + lData2[t + HALF - 8];
        for (int t = 0; t < WG; t++) lData2[t + HALF] = lData[t + HALF] + lData[t + HALF - 16];
        for (int t = 0; t < WG; t++) lData[t + HALF] = lData2[t + HALF] + lData2[t + HALF - 32];
        for (int t = 0; t < WG; t++) {
            const int col = it * WG + t;
            if (col < cols) out[col] = lData[t + HALF] + prevMaxVal;
        }
        prevMaxVal += lData[WG + HALF - 1];
    }
}

// ccl_surf.cpp:451-542 computeIntegralImage: scan rows, transpose, scan rows, transpose back
void integral_image(Surf& S)
{
    const int w = S.W, h = S.H;
    std::vector<float> a((size_t)w * h), t1((size_t)w * h), t2((size_t)w * h);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) scan_row(&S.gray[(size_t)y * w], &a[(size_t)y * w], w);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) t1[(size_t)x * h + y] = a[(size_t)y * w + x];
#pragma omp parallel for schedule(static)
    for (int x = 0; x < w; x++) scan_row(&t1[(size_t)x * h], &t2[(size_t)x * h], h);
    S.integral.resize((size_t)w * h);
    for (int x = 0; x < w; x++)
        for (int y = 0; y < h; y++) S.integral[(size_t)y * w + x] = t2[(size_t)x * h + y];
}

// hessianDet_kernel.cl:77-103 (identical copies in the orientation and descriptor files)
float box_integral(const float* data, int width, int height, int row, int col, int rows, int cols)
{
    float A = 0.0f, B = 0.0f, C = 0.0f, D = 0.0f;
    const int r1 = std::min(row, height) - 1;
    const int c1 = std::min(col, width) - 1;
    const int r2 = std::min(row + rows, height) - 1;
    const int c2 = std::min(col + cols, width) - 1;
    if (r1 >= 0 && c1 >= 0) A = data[r1 * width + c1];
    if (r1 >= 0 && c2 >= 0) B = data[r1 * width + c2];
    if (r2 >= 0 && c1 >= 0) C = data[r2 * width + c1];
    if (r2 >= 0 && c2 >= 0) D = data[r2 * width + c2];
    return fmaxf(0.0f, A - B - C + D);
}

// getOrientation_kernels.cl:113-134
float haar_x(const float* img, int w, int h, int row, int column, int s)
{
    return box_integral(img, w, h, row - s / 2, column, s, s / 2) -
           box_integral(img, w, h, row - s / 2, column - s / 2, s, s / 2);
}

float haar_y(const float* img, int w, int h, int row, int column, int s)
{
    return box_integral(img, w, h, row, column - s / 2, s / 2, s) -
           box_integral(img, w, h, row - s / 2, column - s / 2, s / 2, s);
}

// atan of a non-negative (possibly infinite or NaN) argument
float atan_pos(float v)
{
    if (v != v) return v;
    if (v > 1.0f) return MIF_PIO2_F - mif_atan01(1.0f / v);
    return mif_atan01(v);
}

// getOrientation_kernels.cl:138-153
float get_angle(float X, float Y)
{
    if (X >= 0 && Y >= 0) return atan_pos(Y / X);
    if (X < 0 && Y >= 0) return PI_F - atan_pos(-Y / X);
    if (X < 0 && Y < 0) return PI_F + atan_pos(Y / X);
    if (X >= 0 && Y < 0) return 2 * PI_F - atan_pos(-Y / X);
    return 0;
}

// getOrientation_kernels.cl:68-71
float gaussian(float x, float y, float sig)
{
    return (1.0f / (2.0f * PI_F * sig * sig)) * mif_expf(-(x * x + y * y) / (2.0f * sig * sig));
}

// hessianDet_kernel.cl:105-175 for one layer
void hessian_layer(const Surf& S, Layer& L)
{
    const float* img = S.integral.data();
    const int width = S.W, height = S.H;
#pragma omp parallel for schedule(static)
    for (int idy = 0; idy < L.h; idy++)
        for (int idx = 0; idx < L.w; idx++) {
            const int w = L.filter;
            const int l = L.filter / 3;
            const int b = (L.filter - 1) / 2 + 1;
            const float inverse_area = 1.0f / (w * w);
            const int r = idy * L.step;
            const int c = idx * L.step;
            if (r >= height || c >= width) continue;
            float Dxx = box_integral(img, width, height, r - l + 1, c - b, 2 * l - 1, w) -
                        box_integral(img, width, height, r - l + 1, c - l / 2, 2 * l - 1, l) * 3;
            float Dyy = box_integral(img, width, height, r - b, c - l + 1, w, 2 * l - 1) -
                        box_integral(img, width, height, r - l / 2, c - l + 1, l, 2 * l - 1) * 3;
            float Dxy = box_integral(img, width, height, r - l, c + 1, l, l) +
                        box_integral(img, width, height, r + 1, c - l, l, l) -
                        box_integral(img, width, height, r - l, c - l, l, l) -
                        box_integral(img, width, height, r + 1, c + 1, l, l);
            Dxx *= inverse_area;
            Dyy *= inverse_area;
            Dxy *= inverse_area;
            L.resp[idy * L.w + idx] = (Dxx * Dyy - 0.81f * Dxy * Dxy);
            L.lap[idy * L.w + idx] = (Dxx + Dyy >= 0 ? 1 : 0);
        }
}

inline float resp_at(const Layer& L, int c, int r, int scale) { return L.resp[r * scale * L.w + c * scale]; }

// nonMaxSuppression_kernel.cl:232-296 isExtremum
bool is_extremum(const Layer& t, const Layer& m, const Layer& b, int c, int r, float threshold)
{
    const int layerBorder = (t.filter + 1) / (2 * t.step);
    if (r <= layerBorder || r >= t.h - layerBorder || c <= layerBorder || c >= t.w - layerBorder) return false;
    const int mScale = m.w / t.w;
    const float candidate = resp_at(m, c, r, mScale);
    if (candidate < threshold) return false;
    float localMax = resp_at(t, c - 1, r - 1, 1);
    for (int dr = -1; dr <= 1; dr++)
        for (int dc = -1; dc <= 1; dc++) localMax = fmaxf(localMax, resp_at(t, c + dc, r + dr, 1));
    const int bScale = b.w / t.w;
    for (int dr = -1; dr <= 1; dr++)
        for (int dc = -1; dc <= 1; dc++) localMax = fmaxf(localMax, resp_at(b, c + dc, r + dr, bScale));
    for (int dr = -1; dr <= 1; dr++)
        for (int dc = -1; dc <= 1; dc++)
            if (dr != 0 || dc != 0) localMax = fmaxf(localMax, resp_at(m, c + dc, r + dr, mScale));
    if (localMax > candidate) return false;
    return true;
}

// nonMaxSuppression_kernel.cl:107-228 interpolateExtremum
bool interpolate_extremum(const Layer& t, const Layer& m, const Layer& b, int r, int c, Ipt* out)
{
    const int mScale = m.w / t.w;
    const int bScale = b.w / t.w;
    const float dx = (resp_at(m, c + 1, r, mScale) - resp_at(m, c - 1, r, mScale)) / 2.0f;
    const float dy = (resp_at(m, c, r + 1, mScale) - resp_at(m, c, r - 1, mScale)) / 2.0f;
    const float ds = (resp_at(t, c, r, 1) - resp_at(b, c, r, bScale)) / 2.0f;
    const float v = resp_at(m, c, r, mScale);
    const float dxx = resp_at(m, c + 1, r, mScale) + resp_at(m, c - 1, r, mScale) - 2.0f * v;
    const float dyy = resp_at(m, c, r + 1, mScale) + resp_at(m, c, r - 1, mScale) - 2.0f * v;
    const float dss = resp_at(t, c, r, 1) + resp_at(b, c, r, bScale) - 2.0f * v;
    const float dxy = (resp_at(m, c + 1, r + 1, mScale) - resp_at(m, c - 1, r + 1, mScale) -
                       resp_at(m, c + 1, r - 1, mScale) + resp_at(m, c - 1, r - 1, mScale)) / 4.0f;
    const float dxs = (resp_at(t, c + 1, r, 1) - resp_at(t, c - 1, r, 1) - resp_at(b, c + 1, r, bScale) +
                       resp_at(b, c - 1, r, bScale)) / 4.0f;
    const float dys = (resp_at(t, c, r + 1, 1) - resp_at(t, c, r - 1, 1) - resp_at(b, c, r + 1, bScale) +
                       resp_at(b, c, r - 1, bScale)) / 4.0f;
    const float H0 = dxx, H1 = dxy, H2 = dxs, H3 = dxy, H4 = dyy, H5 = dys, H6 = dxs, H7 = dys, H8 = dss;
    const float determinant = H0 * (H4 * H8 - H7 * H5) - H1 * (H3 * H8 - H5 * H6) + H2 * (H3 * H7 - H4 * H6);
    const float invdet = 1.0f / determinant;
    const float invH0 = (H4 * H8 - H7 * H5) * invdet;
    const float invH1 = -(H3 * H8 - H5 * H6) * invdet;
    const float invH2 = (H3 * H7 - H6 * H4) * invdet;
    const float invH3 = -(H1 * H8 - H2 * H7) * invdet;
    const float invH4 = (H0 * H8 - H2 * H6) * invdet;
    const float invH5 = -(H0 * H7 - H6 * H1) * invdet;
    const float invH6 = (H1 * H5 - H2 * H4) * invdet;
    const float invH7 = -(H0 * H5 - H3 * H2) * invdet;
    const float invH8 = (H0 * H4 - H3 * H1) * invdet;
    float xi, xr, xc;
    xc = invH0 * dx * -1.0f;
    xc += invH1 * dy * -1.0f;
    xc += invH2 * ds * -1.0f;
    xr = invH3 * dx * -1.0f;
    xr += invH4 * dy * -1.0f;
    xr += invH5 * ds * -1.0f;
    xi = invH6 * dx * -1.0f;
    xi += invH7 * dy * -1.0f;
    xi += invH8 * ds * -1.0f;
    if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) {
        const int filterStep = m.filter - b.filter;
        out->x = (float)(((float)c + xc) * (float)t.step);
        out->y = (float)(((float)r + xr) * (float)t.step);
        out->scale = (float)(0.1333f) * ((float)m.filter + (xi * (float)filterStep));
        const int scale = m.w / t.w;
        out->laplacian = m.lap[(r * scale) * m.w + c * scale];
        return true;
    }
    return false;
}

// getOrientation_kernels.cl:161-316 Step1 + Step2 for one interest point
float orientation_one(const Surf& S, const Ipt& p)
{
    const float* img = S.integral.data();
    const int s = (int)mif_roundf(p.scale);
    const int r = (int)mif_roundf(p.y);
    const int c = (int)mif_roundf(p.x);
    float rx[109], ry[109], rz[109];
    int n = 0;
    for (int localId = 0; localId < 169; localId++) {
        const int i = (int)(localId / 13) - 6;
        const int j = (localId % 13) - 6;
        if (i * i + j * j < 36) {
            const float gauss = GAUSS25[7 * TAB_ID[i + 6] + TAB_ID[j + 6]];
            rx[n] = gauss * haar_x(img, S.W, S.H, r + j * s, c + i * s, 4 * s);
            ry[n] = gauss * haar_y(img, S.W, S.H, r + j * s, c + i * s, 4 * s);
            rz[n] = get_angle(rx[n], ry[n]);
            n++;
        }
    }
    float sumx[42], sumy[42];
    for (int tid = 0; tid < 42; tid++) {
        const float ang1 = 0.15f * (float)tid;
        const float ang2 = (ang1 + (PI_F / 3.0f) > 2 * PI_F ? ang1 - (5.0f * PI_F / 3.0f) : ang1 + (PI_F / 3.0f));
        float sx = 0.f, sy = 0.f;
        if (tid <= 34) {
            for (int k = 0; k < 109; ++k) {
                const float ang = rz[k];
                const int check = ang1 < ang && ang < ang2;
                sx += rx[k] * (float)check;
                sy += ry[k] * (float)check;
            }
        } else {
            for (int k = 0; k < 109; ++k) {
                const float ang = rz[k];
                const int check = ((ang > 0 && ang < ang2) || (ang > ang1 && ang < 2 * PI_F));
                sx += rx[k] * (float)check;
                sy += ry[k] * (float)check;
            }
        }
        sumx[tid] = sx;
        sumy[tid] = sy;
    }
    for (int offset = 32; offset > 0; offset >>= 1)
        for (int tid = 0; tid < offset && tid + offset < 42; tid++) {
            const float m0 = sumx[tid] * sumx[tid] + sumy[tid] * sumy[tid];
            const float m1 = sumx[tid + offset] * sumx[tid + offset] + sumy[tid + offset] * sumy[tid + offset];
            if (m0 < m1) {
                sumx[tid] = sumx[tid + offset];
                sumy[tid] = sumy[tid + offset];
            }
        }
    return get_angle(sumx[0], sumy[0]);
}

// createDescriptors_kernel.cl:160-275 + normalizeDescriptors_kernel.cl:54-111
void descriptor_one(const Surf& S, Ipt& p)
{
    const float* img = S.integral.data();
    float lengths[16];
    const float x = mif_roundf(p.x);
    const float y = mif_roundf(p.y);
    const float orient = p.orientation;
    float si, co;
    mif_sincosf(orient, &si, &co);
    const float thScale = p.scale;
    for (int bIdx = 0; bIdx < 16; bIdx++) {
        float cx = 0.5f, cy = 0.5f;
        cx += (float)((int)(bIdx / 4));
        cy += (float)((int)(bIdx % 4));
        const int j = TAB_J[bIdx];
        const int i = TAB_I[bIdx];
        const float ix = (float)(i + 5);
        const float jx = (float)(j + 5);
        const float xs = mif_roundf(x + (-jx * thScale * si + ix * thScale * co));
        const float ys = mif_roundf(y + (jx * thScale * co + ix * thScale * si));
        float d[81][4];
        for (int tha = 0; tha < 81; tha++) {
            const int k = i + (tha / 9);
            const int l = j + (tha % 9);
            const int sample_x = (int)mif_roundf(x + ((float)(-l) * thScale * si + (float)k * thScale * co));
            const int sample_y = (int)mif_roundf(y + ((float)l * thScale * co + (float)k * thScale * si));
            const float gauss_s1 = gaussian((float)(xs - (float)sample_x), (float)(ys - (float)sample_y), 2.5f * thScale);
            const int hs = (int)(2 * mif_roundf(thScale));
            const float rx = haar_x(img, S.W, S.H, sample_y, sample_x, hs);
            const float ry = haar_y(img, S.W, S.H, sample_y, sample_x, hs);
            const float rrx = gauss_s1 * (-rx * si + ry * co);
            const float rry = gauss_s1 * (rx * co + ry * si);
            d[tha][0] = rrx;
            d[tha][1] = rry;
            d[tha][2] = fabsf(rrx);
            d[tha][3] = fabsf(rry);
        }
        // sumDesc, createDescriptors_kernel.cl:135-151
        for (int tha = 0; tha < 81 - 64; tha++)
            for (int q = 0; q < 4; q++) d[tha][q] += d[tha + 64][q];
        for (int stride = 32; stride > 0; stride >>= 1)
            for (int tha = 0; tha < stride; tha++)
                for (int q = 0; q < 4; q++) d[tha][q] += d[tha + stride][q];
        const float gauss_s2 = gaussian(cx - 2.0f, cy - 2.0f, 1.5f);
        for (int q = 0; q < 4; q++) d[0][q] *= gauss_s2;
        for (int q = 0; q < 4; q++) p.desc[bIdx * 4 + q] = d[0][q];
        lengths[bIdx] = d[0][0] * d[0][0] + d[0][1] * d[0][1] + d[0][2] * d[0][2] + d[0][3] * d[0][3];
    }
    for (int i = 8; i > 0; i >>= 1)
        for (int tid = 0; tid < i; tid++) lengths[tid] += lengths[tid + i];
    const float lengthOfDescriptor = 1.0f / sqrtf(lengths[0]);
    for (int q = 0; q < 64; q++) p.desc[q] *= lengthOfDescriptor;
}

} // namespace

extern "C" {

void* orc_surf_create(int width, int height)
{
    if (width <= 0 || height <= 0) return nullptr;
    Surf* S = new Surf();
    S->W = width; S->H = height;
    // ccl_surf.cpp:115-153 createResponseMap for OCTAVES = 3, SAMPLE_STEP = 5
    const int w = width / SAMPLE_STEP, h = height / SAMPLE_STEP, s = SAMPLE_STEP;
    const int spec[8][4] = {{w, h, s, 9}, {w, h, s, 15}, {w, h, s, 21}, {w, h, s, 27},
                            {w / 2, h / 2, s * 2, 39}, {w / 2, h / 2, s * 2, 51},
                            {w / 4, h / 4, s * 4, 75}, {w / 4, h / 4, s * 4, 99}};
    for (int i = 0; i < 8; i++) {
        Layer L;
        L.w = spec[i][0]; L.h = spec[i][1]; L.step = spec[i][2]; L.filter = spec[i][3];
        L.resp.assign((size_t)std::max(L.w * L.h, 1), 0.0f);
        L.lap.assign((size_t)std::max(L.w * L.h, 1), 0);
        S->layers.push_back(L);
    }
    return S;
}

void orc_surf_destroy(void* h) { delete (Surf*)h; }

// the shared elementary function behind getAngle's atan(|Y/X|), for restatements written elsewhere
float orc_math_atan_pos(float v) { return atan_pos(v); }

// ccl_surf.cpp:687-731 Surf::run + 789-820 ccl_surf::run. Returns the number of
// interest points (0 where the reference returns an empty vector).
int orc_surf_run(void* hdl, const uint8_t* pixels, int channels)
{
    Surf& S = *(Surf*)hdl;
    const int w = S.W, h = S.H;
    S.gray.resize((size_t)w * h);
    for (int i = 0; i < w * h; i++) {
        int g;
        if (channels == 3) {
            const int b = pixels[3 * i], gg = pixels[3 * i + 1], r = pixels[3 * i + 2];
            g = (b * 1868 + gg * 9617 + r * 4899 + (1 << 13)) >> 14;
        } else {
            g = pixels[i];
        }
        S.gray[i] = (float)((double)g * (1.0 / 255.0));
    }
    integral_image(S);
    for (auto& L : S.layers) {
        std::fill(L.resp.begin(), L.resp.end(), 0.0f);
        std::fill(L.lap.begin(), L.lap.end(), 0);
        if (L.w > 0 && L.h > 0) hessian_layer(S, L);
    }
    // ccl_surf.cpp:109-113 filter_map, 285-343 selectIpoints
    const int filter_map[3][4] = {{0, 1, 2, 3}, {1, 3, 4, 5}, {3, 5, 6, 7}};
    S.ipts.clear();
    S.raw_count = 0;
    for (int o = 0; o < OCTAVES; o++)
        for (int i = 0; i <= 1; i++) {
            const Layer& b = S.layers[filter_map[o][i]];
            const Layer& m = S.layers[filter_map[o][i + 1]];
            const Layer& t = S.layers[filter_map[o][i + 2]];
            if (t.w <= 0 || t.h <= 0) continue;
            // the NDRange spans the middle layer, coordinates are read as top-layer ones
            for (int r = 0; r < m.h; r++)
                for (int c = 0; c < m.w; c++) {
                    if (!is_extremum(t, m, b, c, r, THRES)) continue;
                    Ipt p;
                    memset(&p, 0, sizeof(p));
                    if (!interpolate_extremum(t, m, b, r, c, &p)) continue;
                    if (S.raw_count < INIT_FEATS) S.ipts.push_back(p);
                    S.raw_count++;
                }
        }
    if (S.ipts.empty()) return 0;
    const int n = (int)S.ipts.size();
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n; i++) {
        S.ipts[i].orientation = orientation_one(S, S.ipts[i]);
        descriptor_one(S, S.ipts[i]);
    }
    return n;
}

int orc_surf_count(void* h) { return (int)((Surf*)h)->ipts.size(); }
int orc_surf_raw_count(void* h) { return ((Surf*)h)->raw_count; }
const float* orc_surf_integral(void* h) { return ((Surf*)h)->integral.data(); }
int orc_surf_layer_info(void* h, int i, int* w, int* hh, int* step, int* filter)
{
    const Layer& L = ((Surf*)h)->layers[i];
    *w = L.w; *hh = L.h; *step = L.step; *filter = L.filter;
    return 0;
}
const float* orc_surf_layer_responses(void* h, int i) { return ((Surf*)h)->layers[i].resp.data(); }
const int* orc_surf_layer_laplacians(void* h, int i) { return ((Surf*)h)->layers[i].lap.data(); }
// rows of 70 floats: x, y, scale, orientation, laplacian (as float), 0, descriptor[64]
void orc_surf_points(void* h, float* out)
{
    const Surf& S = *(Surf*)h;
    for (size_t i = 0; i < S.ipts.size(); i++) {
        const Ipt& p = S.ipts[i];
        float* o = out + i * 70;
        o[0] = p.x; o[1] = p.y; o[2] = p.scale; o[3] = p.orientation; o[4] = (float)p.laplacian; o[5] = 0;
        memcpy(o + 6, p.desc, sizeof(p.desc));
    }
}

} // extern "C"
