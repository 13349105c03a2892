// GPU-resident SIFT for MI355X (gfx950): Gaussian scale space, DoG, extremum
// detection with sub-pixel refinement, orientation and 128-d descriptors.
//
// Replaces the reference's OpenCL kernels and their launch plan
// (SIFT/sift.cpp:41-766 kernels, 1119-1282 call_async_detector / call_async,
// 1856-1916 extract_features) with one stream-ordered sequence per frame that
// never returns to the host between stages:
//
//   6 x blur_level per octave (H + V in one pass through LDS; level 0 also forms its input,
//        the gray conversion of the frame resp. the decimated level 3 of the previous octave)
//        -> detect (extrema of the DoG stack, formed on the fly, as a bitmask; gradient planes)
//        -> refine -> scan -> emit (raster order) -> orientation -> descriptor + normalise
//
// Arithmetic contract: every value is computed with the same sequence of IEEE
// fp32/fp64 operations on every run and every machine (-ffp-contract=off, fused
// operations spelled fmaf, transcendental functions from mif_math.h), so
// keypoint counts, positions, orientations and descriptors are reproducible bit
// for bit and independent of scheduling.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include "mif_common.h"
#include "mif_math.h"

namespace {

constexpr int kOctaves = MIF_SIFT_OCTAVES;
constexpr int kLevels = MIF_SIFT_LEVELS;
constexpr int kIntervals = 3;      // _SIFT_INTVLS, sift.cpp:17
constexpr int kBorder = 5;         // _SIFT_IMG_BORDER, sift.cpp:22
constexpr int kInterpSteps = 5;    // _SIFT_MAX_INTERP_STEPS, sift.cpp:18
constexpr int kOriBins = 36;       // _SIFT_ORI_HIST_BINS, sift.cpp:26
constexpr int kOriLanes = 16;      // _SIFT_ORIENT_GPU_BLOCKSIZE, sift.cpp:27
constexpr int kMaxTaps = 24;
constexpr float kResampleRatio = 2.0f;   // _resample_ratio_x / _y, sift.cpp:812-813
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kBlurTile = 64;      // output tile edge of the blur kernel
// Blocks per frame of the descriptor and orientation kernels (their grids' x extent). The
// lists' capacity is about ten times what a frame yields, so a grid of one block per slot would
// mostly start blocks that exit at once; a block walks the list with the grid's stride instead.
// Few blocks per frame balance badly (keypoints differ in cost): both values were measured.
// Small batches get more blocks per frame, so that the launch still fills the GPU.
constexpr int kDescBlocks = 1536, kDescBlocksTotal = 65536;
constexpr int kOriBlocks = 128, kOriBlocksTotal = 32768;
#ifndef MIF_ORI_UNROLL
#define MIF_ORI_UNROLL 4
#endif
constexpr int kOriUnroll = MIF_ORI_UNROLL;   // window pixels a lane of k_orientation handles per turn

// Kernel constants as the reference's kernels see them: the host prints them
// into -D options with 6 significant digits and OpenCL C parses them as double
// literals (sift.cpp:1372-1394).
#define K_SIGMA 1.6
#define K_CONTR_THR_1 0.0108333
#define K_CONTR_THR 0.065
#define K_ORI_SIG_FCTR 1.5
#define K_ORI_RADIUS 4.5
#define K_ORI_PEAK_RATIO 0.8
#define K_DESCR_MAG_THR 0.2
#define K_INT_DESCR_FCTR 512.0
#define K_M_PI 3.14159265358979323846
#define K_M_PI2 (2.0 * K_M_PI)
#define K_RPI (4.0 / K_M_PI)

struct Taps {
    float t[kMaxTaps];
};

// Thresholds of the orientation histogram: bin(theta) >= b exactly when theta >= t[b]
// (t[0] = -inf, t[kOriBins + 1] = +inf), see build_ori_bins.
struct OriBins {
    float t[kOriBins + 2];
};

// counts[] layout in device memory
enum { CNT_OCT0 = 0, CNT_TOTAL = 3, CNT_RAW0 = 4, CNT_STATUS = 7, CNT_WORDS = 8 };

// ---------------------------------------------------------------- base image

// Consecutive blocks of a launch go to consecutive XCDs (8 of them, each with its own L2).
// The bandwidth kernels below run on a 1-D grid of 8 * ceil(nitems / 8) blocks and renumber
// it so that every XCD owns ONE contiguous range of the work items (tiles or strips in
// frame-major raster order): neighbours, which share halo rows and columns, then meet in the
// same L2 instead of fetching the halo from memory once per XCD. Returns -1 for a surplus block.
__device__ __forceinline__ int xcd_contiguous_item(int nitems)
{
    const int per_xcd = (nitems + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int t = xcd * per_xcd + (int)(blockIdx.x >> 3);
    return (t < min((xcd + 1) * per_xcd, nitems)) ? t : -1;
}

// sift.cpp:901-913 copy_pixels_1ch8_to_1ch32 (a host loop in the reference)
// All kernels below take the frame of a batch from the last grid dimension and
// address per-frame data at frame * stride.
__global__ void k_gray_u8(const uint8_t* __restrict__ in, float* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t f = blockIdx.z;
    if (i < n) out[f * n + i] = (float)in[f * n + i] / 255.0f;
}

// the same, four pixels per thread (one 32-bit word in, one float4 out); n % 4 == 0
__global__ void k_gray_u8_x4(const uint8_t* __restrict__ in, float* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // group of four pixels
    const size_t f = blockIdx.z;
    if (i * 4 >= n) return;
    const uint32_t v = reinterpret_cast<const uint32_t*>(in + f * n)[i];
    float4 o;
    o.x = (float)(v & 0xffu) / 255.0f;
    o.y = (float)((v >> 8) & 0xffu) / 255.0f;
    o.z = (float)((v >> 16) & 0xffu) / 255.0f;
    o.w = (float)(v >> 24) / 255.0f;
    reinterpret_cast<float4*>(out + f * n)[i] = o;
}

// sift.cpp:55-72 kColor_interleaved_ch
__device__ __forceinline__ float gray_of_bgr(float b, float g, float r)
{
    const float cb = (float)0.114, cg = (float)0.587, cr = (float)0.299;
    const float val_scale = (float)0.0039215686;
    return (b * cb + g * cg + r * cr) * val_scale;
}

__global__ void k_gray_bgr(const uint8_t* __restrict__ in, float* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    in += (size_t)blockIdx.z * n * 3;
    out += (size_t)blockIdx.z * n;
    out[i] = gray_of_bgr((float)in[3 * i], (float)in[3 * i + 1], (float)in[3 * i + 2]);
}

// Same conversion, four pixels per thread: 12 bytes come in as three 32-bit
// words and a float4 goes out. Needs n % 4 == 0 and a 4-byte aligned frame.
__global__ void k_gray_bgr_x4(const uint8_t* __restrict__ in, float* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // group of four pixels
    if (4 * i >= n) return;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(in + (size_t)blockIdx.z * n * 3) + 3 * (size_t)i;
    const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];      // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
    float4 o;
    o.x = gray_of_bgr((float)(w0 & 255u), (float)((w0 >> 8) & 255u), (float)((w0 >> 16) & 255u));
    o.y = gray_of_bgr((float)(w0 >> 24), (float)(w1 & 255u), (float)((w1 >> 8) & 255u));
    o.z = gray_of_bgr((float)((w1 >> 16) & 255u), (float)(w1 >> 24), (float)(w2 & 255u));
    o.w = gray_of_bgr((float)((w2 >> 8) & 255u), (float)((w2 >> 16) & 255u), (float)(w2 >> 24));
    reinterpret_cast<float4*>(out + (size_t)blockIdx.z * n)[i] = o;
}

// First octave at twice the input size (up_scale): kResize_NN_3ch_8b followed by
// kColor_interleaved_ch, resp. the host conversion followed by kResize_NN_1ch_32b
// (sift.cpp:134-210, 1140-1177), in one pass. (w, h) is the resampled size, (ws, hs)
// the input's; pixels whose source falls outside are 0.
template <int C>
__global__ void k_base_upscaled(const uint8_t* __restrict__ in, float* __restrict__ out, int w, int h, int ws,
                                int hs, float ratio)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= h) return;
    in += (size_t)blockIdx.z * ws * hs * C;
    out += (size_t)blockIdx.z * w * h;
    const float fy = ((float)y + 0.4995f) / ratio;
    const float fx = ((float)x + 0.4995f) / ratio;
    const int ps = (int)fy * ws + (int)fx;
    const bool inside = fx < (float)ws && fy < (float)hs;
    float v;
    if (C == 3) {
        const float b = inside ? (float)in[3 * ps] : 0.0f;
        const float g = inside ? (float)in[3 * ps + 1] : 0.0f;
        const float r = inside ? (float)in[3 * ps + 2] : 0.0f;
        v = gray_of_bgr(b, g, r);
    } else {
        v = inside ? (float)in[ps] / 255.0f : 0.0f;
    }
    out[(size_t)y * w + x] = v;
}

// sift.cpp:136-208 kResize_NN_1ch_32b / kResize_NN_1ch_8b / kResize_NN_3ch_8b for the probes
// test_kResize and test_ColorToGray_Resize (sift.cpp:2097-2144): nearest-neighbour resampling by
// (ratio_x, ratio_y) of an image of C interleaved channels; destination pixels whose source position
// falls outside the image are 0.
template <typename T, int C>
__global__ void k_resize_nn(T* __restrict__ dst, const T* __restrict__ src, int wd, int hd, int ws, int hs,
                            float ratio_x, float ratio_y)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x > wd - 1 || y > hd - 1) return;
    const float fy = ((float)y + 0.4995f) / ratio_y;
    const float fx = ((float)x + 0.4995f) / ratio_x;
    const int pd = y * wd + x;
    const int ps = (int)fy * ws + (int)fx;
    const bool inside = fx < (float)ws && fy < (float)hs;
#pragma unroll
    for (int c = 0; c < C; c++) dst[(size_t)pd * C + c] = inside ? src[(size_t)ps * C + c] : (T)0;
}

// sift.cpp:113-132 kDown_pyramid with wo = 2w, d = 2, ss_img_id = 3
// (sift.cpp:1187-1191). `src` is the previous octave's whole pyramid; the
// source offset 3*(2w)*(2h) and row stride 2w are kept as written, which differs
// from the previous octave's true geometry when its width or height is odd.
__global__ void k_decimate(const float* __restrict__ src, size_t src_stride, float* __restrict__ dst, int w, int h)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= h) return;
    src += (size_t)blockIdx.z * src_stride;
    dst += (size_t)blockIdx.z * w * h;
    const int wo = 2 * w;
    const int offset = kIntervals * w * 2 * h * 2;
    dst[y * w + x] = src[(y * 2) * wo + x * 2 + offset];
}

// ---------------------------------------------------------------- blur level
//
// One launch per pyramid level: dst = V(H(src)) with the level's taps, i.e.
// kBlurH + kBlurV of the reference (sift.cpp:213-245) in one pass over HBM:
// 4 B read + 4 B written per pixel instead of 8 + 8. The DoG images of kDiff
// (sift.cpp:247-262) are not stored at all: the detector and the emitter, their
// only readers, subtract neighbouring levels where they need a DoG sample.
//
// Both 1-D passes apply taps 0 .. 2*FR-1 only (the reference's loops stop one
// short of the filter width, sift.cpp:222,238) in ascending order as an fmaf
// chain starting from 0, with zeros outside the image.
//
// A block of 256 threads owns a 64 x 64 output tile. Its speed depends on how
// many blocks a CU holds (the memory phases of some hide behind the arithmetic
// of others) and on the instructions spent per pixel, so
//  * ONE LDS array holds first the (64+2FR-1)^2 input window, transposed, and
//    then the horizontal pass's result: every thread moves its horizontal
//    window into registers before the first result is written;
//  * staging: a wave takes one row of the window per step (row index and bounds
//    are wave-uniform, lanes run along x) with a scalar row base, issues all
//    its loads before the first LDS store, and stores TRANSPOSED (odd pitch);
//    rows outside the image are read from a row of zeros;
//  * horizontal pass: lanes run along y, a thread owns one row and 22 columns,
//    so that all four waves carry the same load;
//  * vertical pass: a thread owns two adjacent columns and 8 rows and evaluates
//    the two columns' chains side by side with packed FMAs (each half of a
//    packed FMA is an IEEE fmaf, so the chains are the same as the scalar ones).
// All LDS accesses of the passes have unit or odd lane stride.
// Tiles (frame-major, then raster order) are cut into 8 contiguous ranges, one
// per XCD (consecutive blocks go to consecutive XCDs), so that neighbouring
// tiles, which share halo rows and columns, meet in the same L2.

// Where a blur launch takes its input from. Level 0 of an octave blurs the octave's base image;
// that image is a pure function of the frame's pixels (octave 0) or of the previous octave's level
// 3 (octaves 1, 2), so the level-0 launch forms it while staging its window and the base image is
// never written to memory nor read back:
//   kSrcFloat     a float image [h][w] (levels 1..5, and level 0 when the base image is stored)
//   kSrcGrayU8    8-bit gray frame, value / 255 (sift.cpp:901-913)
//   kSrcBgrU8     8-bit interleaved BGR frame, kColor_interleaved_ch (sift.cpp:55-72)
//   kSrcDecimated previous octave's pyramid, kDown_pyramid (sift.cpp:113-132) with its flat
//                 indexing: pixel (x, y) = prev[3*(2w)*(2h) + (2y)*(2w) + 2x]
enum { kSrcFloat = 0, kSrcGrayU8 = 1, kSrcBgrU8 = 2, kSrcDecimated = 3 };

// Stages the EW x EH window of the source whose top-left pixel is (wx0, wy0) into LDS, transposed:
// s_in[column * PT + row]. A wave takes one row of the window per step (row index and bounds are
// wave-uniform, lanes run along x) with a scalar row base and issues all its loads before the first LDS
// store; rows outside the image are read from a row of zeros, columns outside become zeros. `src` is the
// frame's source as SRC describes it (see the enum above). 64 <= EW <= 128.
template <int SRC, int EW, int EH, int PT>
__device__ __forceinline__ void stage_window(const char* __restrict__ src, int w, int h, int wx0, int wy0, int lane,
                                             int wv, const float* __restrict__ zero_row, float* __restrict__ s_in)
{
    constexpr int kSteps = (EH + 3) / 4;         // staged rows per wave
    static_assert(EW >= 64 && EW - 64 <= 64, "window width");
    const int gx0 = wx0 + lane, gx1 = gx0 + 64;
    const bool in0 = gx0 >= 0 && gx0 < w;
    const bool in1 = gx1 >= 0 && gx1 < w;
    // the second load covers the columns beyond 64; its idle lanes repeat the last one
    const int cx0 = min(max(gx0, 0), w - 1);
    const int cx1 = min(max(min(gx1, wx0 + EW - 1), 0), w - 1);
    // byte offset of a lane's pixel within its row, and the row pitch in bytes. A BGR pixel is
    // fetched as one (unaligned) 32-bit word that starts at its B byte, or one byte earlier for
    // the last pixel of a row so that the word ends with the row.
    constexpr unsigned px_bytes = SRC == kSrcFloat ? 4u : SRC == kSrcGrayU8 ? 1u : SRC == kSrcBgrU8 ? 3u : 8u;
    const unsigned row_bytes = (SRC == kSrcDecimated ? 4u * 4u : px_bytes) * (unsigned)w;
    const unsigned lim = 3u * (unsigned)w - 4u;
    const unsigned boff0 = (SRC == kSrcBgrU8) ? min(px_bytes * (unsigned)cx0, lim) : px_bytes * (unsigned)cx0;
    const unsigned boff1 = (SRC == kSrcBgrU8) ? min(px_bytes * (unsigned)cx1, lim) : px_bytes * (unsigned)cx1;
    const unsigned sh0 = 8u * (3u * (unsigned)cx0 - boff0), sh1 = 8u * (3u * (unsigned)cx1 - boff1);
    const char* const img0 = src + (SRC == kSrcDecimated ? (size_t)4 * kIntervals * (size_t)(w * 2) * (size_t)(h * 2) : (size_t)0);
    auto fetch = [&](const char* at) -> float {
        if (SRC == kSrcFloat || SRC == kSrcDecimated) return *(const float*)at;
        uint32_t v;
        if (SRC == kSrcGrayU8) v = *(const uint8_t*)at;
        else __builtin_memcpy(&v, at, 4);
        return __builtin_bit_cast(float, v);
    };
    // raw word -> pixel value (the conversions of the reference, see the kernels of the same name)
    auto value = [&](float raw, unsigned sh) -> float {
        if (SRC == kSrcFloat || SRC == kSrcDecimated) return raw;
        const uint32_t v = __builtin_bit_cast(uint32_t, raw);
        if (SRC == kSrcGrayU8) return (float)v / 255.0f;
        const uint32_t u = v >> sh;
        return gray_of_bgr((float)(u & 255u), (float)((u >> 8) & 255u), (float)((u >> 16) & 255u));
    };
    float st0[kSteps], st1[kSteps];
#pragma unroll
    for (int it = 0; it < kSteps; it++) {
        const int gy = wy0 + wv + 4 * it;
        const int gyc = min(max(gy, 0), h - 1);
        const char* inside = img0 + (unsigned)gyc * row_bytes;
        const char* rowp = (gy == gyc) ? inside : (const char*)zero_row;
        st0[it] = fetch(rowp + boff0);
        st1[it] = fetch(rowp + boff1);
    }
#pragma unroll
    for (int it = 0; it < kSteps; it++) {
        const int ly = wv + 4 * it;
        if (4 * it + 3 < EH || ly < EH) s_in[lane * PT + ly] = in0 ? value(st0[it], sh0) : 0.0f;
    }
    if (lane < EW - 64) {
#pragma unroll
        for (int it = 0; it < kSteps; it++) {
            const int ly = wv + 4 * it;
            if (4 * it + 3 < EH || ly < EH) s_in[(64 + lane) * PT + ly] = in1 ? value(st1[it], sh1) : 0.0f;
        }
    }
}

// The same window, staged four pixels per lane and load: for images whose width is a multiple of four
// (and whose frames start on a 4-byte boundary, 16-byte for float sources) a lane fetches the aligned
// group of four pixels it owns with ONE load -- 16 bytes of a float image, 12 bytes of a BGR frame,
// 4 bytes of a gray frame -- instead of one (for BGR: unaligned) load per pixel, which cuts the
// window's load instructions to a fifth. The window is extended on the left by LS columns so that it
// starts at a multiple of four; a group then lies inside the image as a whole or not at all. A wave's
// load covers 8 groups x 8 rows (128 / 96 / 32 contiguous bytes of 8 rows); with the transposed layout
// s_in[column * PT + row] (PT odd) the four stores of such a block put two lanes on every LDS bank, the
// minimum for 64 lanes. All loads of a thread are issued before its first LDS store.
template <int SRC, int EW, int EH, int PT, int LS>
__device__ __forceinline__ void stage_window4(const char* __restrict__ src, int w, int h, int wx0, int wy0, int tid,
                                              float* __restrict__ s_in)
{
    static_assert(SRC == kSrcFloat || SRC == kSrcGrayU8 || SRC == kSrcBgrU8, "sources with four pixels per load");
    constexpr int G = (EW + LS + 3) / 4;                 // groups per window row
    constexpr int GB = (G + 7) / 8, RB = (EH + 7) / 8;   // blocks of 8 groups x 8 rows
    constexpr int NBLK = GB * RB;
    constexpr int ITER = (NBLK + 3) / 4;                 // blocks per wave
    constexpr int NW = SRC == kSrcFloat ? 4 : SRC == kSrcBgrU8 ? 3 : 1;     // 32-bit words per group
    constexpr unsigned px_bytes = SRC == kSrcFloat ? 4u : SRC == kSrcBgrU8 ? 3u : 1u;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = lane & 7, lr = lane >> 3;
    const unsigned row_bytes = px_bytes * (unsigned)w;
    uint32_t raw[ITER][NW];
#pragma unroll
    for (int it = 0; it < ITER; it++) {
        const int blk = min(wv + 4 * it, NBLK - 1);      // wave-uniform; a surplus turn repeats the last block
        const int rb = blk / GB, gb = blk - rb * GB;
        const int gx = wx0 - LS + 4 * (gb * 8 + lg), gy = wy0 + rb * 8 + lr;
        const unsigned off = (unsigned)min(max(gy, 0), h - 1) * row_bytes + (unsigned)min(max(gx, 0), w - 4) * px_bytes;
        const uint32_t* at = (const uint32_t*)(src + off);
#pragma unroll
        for (int k = 0; k < NW; k++) raw[it][k] = at[k];
    }
#pragma unroll
    for (int it = 0; it < ITER; it++) {
        const int blk = wv + 4 * it;
        if (blk >= NBLK) break;                          // wave-uniform
        const int rb = blk / GB, gb = blk - rb * GB;
        const int g = gb * 8 + lg, r = rb * 8 + lr;
        const int gx = wx0 - LS + 4 * g, gy = wy0 + r;
        const bool inside = gx >= 0 && gx < w && gy >= 0 && gy < h;
        float v[4];
        if (SRC == kSrcFloat) {
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = __builtin_bit_cast(float, raw[it][j]);
        } else if (SRC == kSrcGrayU8) {
            const uint32_t w0 = raw[it][0];
            v[0] = (float)(w0 & 255u) / 255.0f;
            v[1] = (float)((w0 >> 8) & 255u) / 255.0f;
            v[2] = (float)((w0 >> 16) & 255u) / 255.0f;
            v[3] = (float)(w0 >> 24) / 255.0f;
        } else {
            const uint32_t w0 = raw[it][0], w1 = raw[it][NW > 1 ? 1 : 0], w2 = raw[it][NW > 2 ? 2 : 0];
            v[0] = gray_of_bgr((float)(w0 & 255u), (float)((w0 >> 8) & 255u), (float)((w0 >> 16) & 255u));
            v[1] = gray_of_bgr((float)(w0 >> 24), (float)(w1 & 255u), (float)((w1 >> 8) & 255u));
            v[2] = gray_of_bgr((float)((w1 >> 16) & 255u), (float)(w1 >> 24), (float)(w2 & 255u));
            v[3] = gray_of_bgr((float)((w2 >> 8) & 255u), (float)((w2 >> 16) & 255u), (float)(w2 >> 24));
        }
        if (r < EH) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int c = 4 * g + j - LS;
                if (c >= 0 && c < EW) s_in[c * PT + r] = inside ? v[j] : 0.0f;
            }
        }
    }
}

template <int FR, int SRC, bool A4>
__global__ __launch_bounds__(256) void k_blur_level(const void* __restrict__ src_any, size_t src_stride,
                                                    float* __restrict__ dst, size_t pyr_stride, int w, int h,
                                                    Taps taps,
                                                    int tiles_x, int tiles_y, int ntiles,
                                                    const float* __restrict__ zero_row)
{
    constexpr int TW = kBlurTile, TH = kBlurTile;
    constexpr int R = 8;                         // rows of a vertical-pass thread
    constexpr int HC = 22, HS = 21;              // columns of a horizontal-pass thread, chunk spacing
    constexpr int NT = 2 * FR;                   // taps applied
    constexpr int EW = TW + NT - 1;              // staged columns
    constexpr int EH = TH + NT - 1;              // staged rows (odd)
    constexpr int PT = EH | 1;                   // odd pitch of the transposed input
    constexpr int PH = TW + 1;                   // odd pitch of the horizontal pass's result
    static_assert(TW == 64 && TH == 64, "thread mappings below assume a 64 x 64 tile");
    static_assert(NT <= kMaxTaps && EW - 64 <= 64, "tap radius too large");
    static_assert(3 * EH <= 256 && 2 * HS + HC == TW, "horizontal pass: three chunks of one row per thread");

    __shared__ float s_buf[(EW * PT > EH * PH) ? EW * PT : EH * PH];
    float* const s_in = s_buf;                   // [column][row] of the input window
    float* const s_h = s_buf;                    // [row][column] after the horizontal pass

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> tile
    const int t = xcd_contiguous_item(ntiles);
    if (t < 0) return;                           // uniform over the block
    const int tiles_xy = tiles_x * tiles_y;
    const int z = t / tiles_xy;
    const int rem = t - z * tiles_xy;
    const int ty = rem / tiles_x;
    const int x0 = (rem - ty * tiles_x) * TW, y0 = ty * TH;
    // src_stride counts elements of the source: floats, or bytes of an 8-bit frame
    const char* const src = (const char*)src_any + (size_t)z * src_stride * ((SRC == kSrcFloat || SRC == kSrcDecimated) ? 4 : 1);
    dst += (size_t)z * pyr_stride;

    if (A4 && SRC != kSrcDecimated) {
        constexpr int LS = (4 - FR % 4) % 4;     // x0 - FR - LS is a multiple of four
        stage_window4<SRC == kSrcDecimated ? kSrcFloat : SRC, EW, EH, PT, LS>(src, w, h, x0 - FR, y0 - FR, tid, s_in);
    } else {
        stage_window<SRC, EW, EH, PT>(src, w, h, x0 - FR, y0 - FR, lane, wv, zero_row, s_in);
    }
    __syncthreads();

    // vertical-pass mapping: columns 2 xp, 2 xp + 1, rows R yc .. R yc + R - 1
    const int xp = tid & 31;
    const int yc = tid >> 5;

    // horizontal pass: chunks start at columns 0, 21, 42; the shared columns are written
    // twice with the same value
    const bool hactive = tid < 3 * EH;
    const int chunk = tid / EH;
    const int row = tid - chunk * EH;
    float hacc[HC];
    {
        float win[HC + NT - 1];
        if (hactive) {
#pragma unroll
            for (int j = 0; j < HC + NT - 1; j++) win[j] = s_in[(chunk * HS + j) * PT + row];
        }
        __syncthreads();                         // the window is in registers everywhere: s_buf is free
        if (hactive) {
#pragma unroll
            for (int p = 0; p < HC; p++) hacc[p] = 0.0f;
#pragma unroll
            for (int k = 0; k < NT; k++) {
#pragma unroll
                for (int p = 0; p < HC; p++) hacc[p] = fmaf(taps.t[k], win[p + k], hacc[p]);
                __builtin_amdgcn_sched_barrier(0);      // keep the chains interleaved
            }
            float* o0 = &s_h[row * PH + chunk * HS];
#pragma unroll
            for (int p = 0; p < HC; p++) o0[p] = hacc[p];
        }
    }
    __syncthreads();

    // vertical pass
    {
        v2f win[R + NT - 1];
#pragma unroll
        for (int j = 0; j < R + NT - 1; j++) {
            const float* p = &s_h[(yc * R + j) * PH + 2 * xp];
            win[j] = (v2f){p[0], p[1]};
        }
        const int gx = x0 + 2 * xp;
        const bool pair = gx + 1 < w && (w & 1) == 0;     // 8-byte aligned two-pixel stores
        v2f acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = (v2f){0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const v2f tk = (v2f){taps.t[k], taps.t[k]};
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = __builtin_elementwise_fma(tk, win[r + k], acc[r]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // pin the results here, or each chain is sunk into its row's bounds check below
#pragma unroll
        for (int r = 0; r < R; r++) asm volatile("" : "+v"(acc[r]));
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int gy = y0 + yc * R + r;
            if (gy < h && gx < w) {
                const size_t at = (size_t)gy * w + gx;
                if (pair) {
                    *(v2f*)&dst[at] = acc[r];
                } else {
                    dst[at] = acc[r].x;
                    if (gx + 1 < w) dst[at + 1] = acc[r].y;
                }
            }
        }
    }
}

// ------------------------------------------------------------ two levels per launch
//
// Levels a and b = a + 1 of an octave from level a - 1 (or, for a = 0, from the octave's source) in
// one launch: level a never comes back from memory, which saves its 4 bytes per pixel of reads, a
// quarter of what two single launches move (r4 + w4 + w4 instead of r4 + w4 + r4 + w4).
//
// A block owns a 64 x 64 tile of level b. It needs level a on the tile extended by level b's taps,
// (64 + 2 FB - 1)^2 pixels, and for that the source on that region extended by level a's taps:
//   stage  source window                          [EH0][EW0], transposed in LDS
//   H_a    horizontal pass with level a's taps     [EH0][OWA]
//   V_a    vertical pass                           [OHA][OWA] = level a on the extended tile; its centre
//          goes to memory as level a, all of it stays in LDS with ZEROS at positions outside the image:
//          level b's passes read level a zero-padded (sift.cpp:222-226, 238-242), not blurred further out
//   H_b    horizontal pass with level b's taps     [OHA][64]
//   V_b    vertical pass                           [64][64]  -> memory as level b
// Every pass is the fmaf chain of k_blur_level (taps 0 .. 2 FR - 1 in ascending order from 0), so
// both levels come out bit for bit as from two single launches. One LDS array serves all phases: every
// thread moves a phase's operands into registers before the phase's results overwrite them.
// The extended tile starts one column further left when FB is odd (`LEAD`), so that the column pairs
// of the vertical pass start at even x and level a is stored with 8-byte stores like level b.
template <int FA, int FB, int SRC, bool A4>
__global__ __launch_bounds__(256) void k_blur_pair(const void* __restrict__ src_any, size_t src_stride,
                                                   float* __restrict__ dst_a, float* __restrict__ dst_b,
                                                   size_t pyr_stride, int w, int h, Taps taps_a, Taps taps_b,
                                                   int tiles_x, int tiles_y, int ntiles,
                                                   const float* __restrict__ zero_row)
{
    constexpr int NTA = 2 * FA, NTB = 2 * FB;
    constexpr int LEAD = FB & 1;
    constexpr int OWA = 64 + NTB - 1 + LEAD;     // level a: columns of the extended tile
    constexpr int OHA = 64 + NTB - 1;            // level a: rows of the extended tile
    constexpr int EW0 = OWA + NTA - 1;           // staged source columns
    constexpr int EH0 = OHA + NTA - 1;           // staged source rows
    constexpr int PT0 = EH0 | 1;                 // odd pitch of the transposed source window
    constexpr int PHA = OWA | 1;                 // odd pitch of H_a's result [row][column]
    constexpr int PA = OWA | 1;                  // odd pitch of level a [row][column]
    constexpr int PHB = 65;                      // odd pitch of H_b's result
    // H_a: a thread owns one row and one of NCH1 column chunks
    constexpr int NCH1 = (3 * EH0 <= 256) ? 3 : 2;
    constexpr int HC1 = (OWA + NCH1 - 1) / NCH1;
    // V_a: a thread owns two adjacent columns and R2 rows
    constexpr int NP2 = (OWA + 1) / 2;
    constexpr int NRC2 = 256 / NP2;
    constexpr int R2 = (OHA + NRC2 - 1) / NRC2;
    // H_b and V_b: as in k_blur_level
    constexpr int HC = 22, HS = 21, R = 8;
    static_assert(NTA <= kMaxTaps && NTB <= kMaxTaps && EW0 - 64 <= 64, "tap radii too large");
    static_assert(NCH1 * EH0 <= 256 && NRC2 * R2 >= OHA && 3 * OHA <= 256, "thread mappings");
    constexpr int c0 = EW0 * PT0, c1 = EH0 * PHA, c2 = OHA * PA, c3 = OHA * PHB;
    constexpr int kFloats = (c0 > c1 ? c0 : c1) > (c2 > c3 ? c2 : c3) ? (c0 > c1 ? c0 : c1) : (c2 > c3 ? c2 : c3);
    __shared__ float s_buf[kFloats];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = xcd_contiguous_item(ntiles);
    if (t < 0) return;                           // uniform over the block
    const int tiles_xy = tiles_x * tiles_y;
    const int z = t / tiles_xy;
    const int rem = t - z * tiles_xy;
    const int ty = rem / tiles_x;
    const int x0 = (rem - ty * tiles_x) * 64, y0 = ty * 64;
    const char* const src = (const char*)src_any + (size_t)z * src_stride * ((SRC == kSrcFloat || SRC == kSrcDecimated) ? 4 : 1);
    dst_a += (size_t)z * pyr_stride;
    dst_b += (size_t)z * pyr_stride;
    const int ax0 = x0 - FB - LEAD, ay0 = y0 - FB;       // image position of the extended tile's corner

    if (A4 && SRC != kSrcDecimated) {
        constexpr int LS = (4 - (FA + FB + LEAD) % 4) % 4;
        stage_window4<SRC == kSrcDecimated ? kSrcFloat : SRC, EW0, EH0, PT0, LS>(src, w, h, ax0 - FA, ay0 - FA, tid, s_buf);
    } else {
        stage_window<SRC, EW0, EH0, PT0>(src, w, h, ax0 - FA, ay0 - FA, lane, wv, zero_row, s_buf);
    }
    __syncthreads();

    // ---- H_a: s_buf[column * PT0 + row] -> s_buf[row * PHA + column]
    {
        const bool active = tid < NCH1 * EH0;
        const int chunk = tid / EH0;
        const int row = tid - chunk * EH0;
        const int cstart = min(chunk * HC1, OWA - HC1);  // the last chunk may overlap its neighbour
        float win[HC1 + NTA - 1];
        if (active) {
#pragma unroll
            for (int j = 0; j < HC1 + NTA - 1; j++) win[j] = s_buf[(cstart + j) * PT0 + row];
        }
        __syncthreads();
        if (active) {
            float acc[HC1];
#pragma unroll
            for (int p = 0; p < HC1; p++) acc[p] = 0.0f;
#pragma unroll
            for (int k = 0; k < NTA; k++) {
#pragma unroll
                for (int p = 0; p < HC1; p++) acc[p] = fmaf(taps_a.t[k], win[p + k], acc[p]);
                __builtin_amdgcn_sched_barrier(0);
            }
            float* o0 = &s_buf[row * PHA + cstart];
#pragma unroll
            for (int p = 0; p < HC1; p++) o0[p] = acc[p];
        }
    }
    __syncthreads();

    // ---- V_a: s_buf[row * PHA + column] -> level a: s_buf[row * PA + column] and memory
    {
        const int xp = tid % NP2;
        const int yc = tid / NP2;
        const bool active = yc < NRC2;
        v2f win[R2 + NTA - 1];
        if (active) {
#pragma unroll
            for (int j = 0; j < R2 + NTA - 1; j++) {
                const float* p = &s_buf[min(yc * R2 + j, EH0 - 1) * PHA + 2 * xp];
                win[j] = (v2f){p[0], p[1]};
            }
        }
        __syncthreads();
        if (active) {
            v2f acc[R2];
#pragma unroll
            for (int r = 0; r < R2; r++) acc[r] = (v2f){0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < NTA; k++) {
                const v2f tk = (v2f){taps_a.t[k], taps_a.t[k]};
#pragma unroll
                for (int r = 0; r < R2; r++) acc[r] = __builtin_elementwise_fma(tk, win[r + k], acc[r]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < R2; r++) asm volatile("" : "+v"(acc[r]));
            const int cx = 2 * xp;
            const int gx = ax0 + cx;
            const bool in_x0 = gx >= 0 && gx < w, in_x1 = gx + 1 >= 0 && gx + 1 < w && cx + 1 < OWA;
            const bool centre_x = cx >= FB + LEAD && cx < FB + LEAD + 64;
            const bool pair = gx + 1 < w && (w & 1) == 0;
#pragma unroll
            for (int r = 0; r < R2; r++) {
                const int cy = yc * R2 + r;
                const int gy = ay0 + cy;
                if (cy < OHA) {
                    const bool in_y = gy >= 0 && gy < h;
                    const float v0 = (in_y && in_x0) ? acc[r].x : 0.0f;
                    const float v1 = (in_y && in_x1) ? acc[r].y : 0.0f;
                    float* q = &s_buf[cy * PA + cx];
                    q[0] = v0;
                    if (cx + 1 < OWA) q[1] = v1;
                    if (centre_x && cy >= FB && cy < FB + 64 && in_y && in_x0) {
                        const size_t at = (size_t)gy * w + gx;
                        if (pair) {
                            *(v2f*)&dst_a[at] = acc[r];
                        } else {
                            dst_a[at] = acc[r].x;
                            if (gx + 1 < w) dst_a[at + 1] = acc[r].y;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- H_b: level a [row][column] -> s_buf[row * PHB + column]
    {
        const bool active = tid < 3 * OHA;
        const int chunk = tid / OHA;
        const int row = tid - chunk * OHA;
        float win[HC + NTB - 1];
        if (active) {
#pragma unroll
            for (int j = 0; j < HC + NTB - 1; j++) win[j] = s_buf[row * PA + LEAD + chunk * HS + j];
        }
        __syncthreads();
        if (active) {
            float acc[HC];
#pragma unroll
            for (int p = 0; p < HC; p++) acc[p] = 0.0f;
#pragma unroll
            for (int k = 0; k < NTB; k++) {
#pragma unroll
                for (int p = 0; p < HC; p++) acc[p] = fmaf(taps_b.t[k], win[p + k], acc[p]);
                __builtin_amdgcn_sched_barrier(0);
            }
            float* o0 = &s_buf[row * PHB + chunk * HS];
#pragma unroll
            for (int p = 0; p < HC; p++) o0[p] = acc[p];
        }
    }
    __syncthreads();

    // ---- V_b: -> memory as level b
    {
        const int xp = tid & 31;
        const int yc = tid >> 5;
        v2f win[R + NTB - 1];
#pragma unroll
        for (int j = 0; j < R + NTB - 1; j++) {
            const float* p = &s_buf[(yc * R + j) * PHB + 2 * xp];
            win[j] = (v2f){p[0], p[1]};
        }
        const int gx = x0 + 2 * xp;
        const bool pair = gx + 1 < w && (w & 1) == 0;
        v2f acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = (v2f){0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < NTB; k++) {
            const v2f tk = (v2f){taps_b.t[k], taps_b.t[k]};
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = __builtin_elementwise_fma(tk, win[r + k], acc[r]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < R; r++) asm volatile("" : "+v"(acc[r]));
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int gy = y0 + yc * R + r;
            if (gy < h && gx < w) {
                const size_t at = (size_t)gy * w + gx;
                if (pair) {
                    *(v2f*)&dst_b[at] = acc[r];
                } else {
                    dst_b[at] = acc[r].x;
                    if (gx + 1 < w) dst_b[at + 1] = acc[r].y;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ detector

// sift.cpp:292-322 LOC_EXTREMUM: the previous image is tested at (y-1,x-1)
// twice and never at (y-1,x+1); the centre of the current image is skipped.
template <bool GT>
__device__ __forceinline__ bool loc_extremum(float v, const float* __restrict__ p, const float* __restrict__ c,
                                             const float* __restrict__ n, int w, int y, int x)
{
#define CMP(img, yy, xx) (GT ? (v > img[(yy) * w + (xx)]) : (v < img[(yy) * w + (xx)]))
    return CMP(p, y, x - 1) && CMP(p, y, x) && CMP(p, y, x + 1) && CMP(p, y + 1, x - 1) && CMP(p, y + 1, x) &&
           CMP(p, y + 1, x + 1) && CMP(p, y - 1, x - 1) && CMP(p, y - 1, x) &&
           CMP(c, y, x - 1) && CMP(c, y, x + 1) && CMP(c, y + 1, x - 1) && CMP(c, y + 1, x) &&
           CMP(c, y + 1, x + 1) && CMP(c, y - 1, x - 1) && CMP(c, y - 1, x) && CMP(c, y - 1, x + 1) &&
           CMP(n, y, x - 1) && CMP(n, y, x) && CMP(n, y, x + 1) && CMP(n, y + 1, x - 1) && CMP(n, y + 1, x) &&
           CMP(n, y + 1, x + 1) && CMP(n, y - 1, x - 1) && CMP(n, y - 1, x) && CMP(n, y - 1, x + 1);
#undef CMP
}

// The 19 DoG samples one Newton step reads around (tx, ty): the 3x3 of the
// current image and the plus-shaped neighbourhoods of the previous and next one.
struct Stencil {
    float c00, cxp, cxm, cyp, cym;        // current: centre, x+1, x-1, y+1, y-1
    float cpp, cmm, cpm, cmp;             // current: (y+1,x+1), (y-1,x-1), (y+1,x-1), (y-1,x+1)
    float n00, nxp, nxm, nyp, nym;        // next image
    float p00, pxp, pxm, pyp, pym;        // previous image
};

struct NewtonStep {
    float Dxx, Dyy, Dxy, gx, gy, gz, ox, oy, oz;
    bool singular;
};

// One iteration of the sub-pixel fit, sift.cpp:339-377: gradient, 3x3 Hessian,
// its adjugate scaled by -1/det, offset = -H^-1 grad. Every operation rounds
// separately in the order written there.
__device__ __forceinline__ NewtonStep newton_step(const Stencil& t)
{
    NewtonStep r;
    r.gx = 0.5f * (t.cxp - t.cxm);
    r.gy = 0.5f * (t.cyp - t.cym);
    r.gz = 0.5f * (t.n00 - t.p00);
    r.Dxx = t.cxp - 2.0f * t.c00 + t.cxm;
    r.Dyy = t.cyp - 2.0f * t.c00 + t.cym;
    const float Dss = t.n00 - 2.0f * t.c00 + t.p00;
    r.Dxy = 0.25f * (t.cpp + t.cmm - t.cpm - t.cmp);
    const float Dxs = 0.25f * (t.nxp + t.pxm - t.nxm - t.pxp);
    const float Dys = 0.25f * (t.nyp + t.pym - t.pyp - t.nym);
    const float Dxx = r.Dxx, Dyy = r.Dyy, Dxy = r.Dxy;
    // adjugate of the 3x3 Hessian, rows as in sift.cpp:356-366
    float a0 = Dyy * Dss - Dys * Dys;
    float b0 = Dys * Dxs - Dxy * Dss;
    float c0 = Dxy * Dys - Dyy * Dxs;
    const float det = Dxx * a0 + Dxy * b0 + Dxs * c0;
    float a1 = Dxs * Dys - Dxy * Dss;
    float a2 = Dxy * Dys - Dxs * Dyy;
    float b1 = Dxx * Dss - Dxs * Dxs;
    float b2 = Dxs * Dxy - Dxx * Dys;
    float c1 = Dxy * Dxs - Dxx * Dys;
    float c2 = Dxx * Dyy - Dxy * Dxy;
    r.singular = (det == 0.0f);
    const float dv = -1.0f / det;
    a0 *= dv; a1 *= dv; a2 *= dv;
    b0 *= dv; b1 *= dv; b2 *= dv;
    c0 *= dv; c1 *= dv; c2 *= dv;
    r.ox = a0 * r.gx + a1 * r.gy + a2 * r.gz;
    r.oy = b0 * r.gx + b1 * r.gy + b2 * r.gz;
    r.oz = c0 * r.gx + c1 * r.gy + c2 * r.gz;
    return r;
}

__device__ __forceinline__ bool step_converged(const NewtonStep& r)
{
    return mif_fabsf(r.ox) < 0.5f && mif_fabsf(r.oy) < 0.5f && mif_fabsf(r.oz) < 0.5f;
}

// sift.cpp:393-415: edge test, contrast test and the feature record, given the
// converged step at (tx, ty, ts) and `cur`, the DoG sample the reference picks
// there (previous / current / next image for ts = 1 / 2 / other).
__device__ __forceinline__ bool finish_pixel(const NewtonStep& r, float cur, int tx, int ty, int ts, int o, float4* out)
{
    const float trH = r.Dxx + r.Dyy;
    const float detH = r.Dxx * r.Dyy - r.Dxy * r.Dxy;
    if ((double)mif_fabsf(detH) <= 1e-8) return false;
    if ((trH * trH) / detH >= 12.0f) return false;    // (10+1)*(10+1)/10 in integers, sift.cpp:398
    const float contrast = cur + 0.5f * (r.gx * r.ox + r.gy * r.oy + r.gz * r.oz);
    if (!(mif_fabsf(contrast) > (float)(K_CONTR_THR / kIntervals))) return false;
    const float p2o = mif_pow2i(o);
    out->x = ((float)tx + r.ox) * p2o;
    out->y = ((float)ty + r.oy) * p2o;
    out->z = (float)(K_SIGMA * (double)mif_exp2f(((float)ts + r.oz) / (float)kIntervals));
    out->w = (float)ts;
    return true;
}

// sift.cpp:323-417, the part of kDetector that follows the extremum test, for
// the pixel (x, y) of DoG level s in octave o: sub-pixel refinement, edge and
// contrast tests. Returns true when the reference would append a feature; *out
// is that feature. The refinement steps tx, ty, ts but keeps reading the same
// three DoG images, as the reference does.
//
// The DoG images are not stored: DoG[d] = level d+1 - level d (kDiff, sift.cpp:256-260) is one float
// subtraction and is evaluated where a sample is needed, from the pyramid `pyr` ([6][h][w] of the frame).
__device__ bool refine_pixel(const float* __restrict__ pyr, size_t items, int w, int h, int s, int o, int x, int y,
                             float4* out)
{
    // previous / current / next DoG image of level s: DoG[s-1], DoG[s], DoG[s+1]
    const float* l0 = pyr + (size_t)(s - 1) * items;
    const float* l1 = l0 + items;
    const float* l2 = l1 + items;
    const float* l3 = l2 + items;
    auto dp = [&](int i) { return l1[i] - l0[i]; };
    auto dc = [&](int i) { return l2[i] - l1[i]; };
    auto dn = [&](int i) { return l3[i] - l2[i]; };
    NewtonStep r;
    int tx = x, ty = y, ts = s, iter;
    for (iter = 0; iter < kInterpSteps; iter++) {
        const int i0 = ty * w + tx;
        Stencil t;
        t.c00 = dc(i0); t.cxp = dc(i0 + 1); t.cxm = dc(i0 - 1); t.cyp = dc(i0 + w); t.cym = dc(i0 - w);
        t.cpp = dc(i0 + w + 1); t.cmm = dc(i0 - w - 1); t.cpm = dc(i0 + w - 1); t.cmp = dc(i0 - w + 1);
        t.n00 = dn(i0); t.nxp = dn(i0 + 1); t.nxm = dn(i0 - 1); t.nyp = dn(i0 + w); t.nym = dn(i0 - w);
        t.p00 = dp(i0); t.pxp = dp(i0 + 1); t.pxm = dp(i0 - 1); t.pyp = dp(i0 + w); t.pym = dp(i0 - w);
        r = newton_step(t);
        if (r.singular) return false;
        if (step_converged(r)) break;
        tx += (int)mif_roundf(r.ox);
        if (tx < kBorder || tx >= w - kBorder) return false;
        ty += (int)mif_roundf(r.oy);
        if (ty < kBorder || ty >= h - kBorder) return false;
        ts += (int)mif_roundf(r.oz);
        if (ts < 1 || ts > kIntervals) return false;
    }
    if (iter >= kInterpSteps) return false;
    const int ic = ty * w + tx;
    const float cur = ts == 1 ? dp(ic) : ts == 2 ? dc(ic) : dn(ic);   // sift.cpp:400
    return finish_pixel(r, cur, tx, ty, ts, o, out);
}

// Detection pass (sift.cpp:276-322), streaming form without a staging tile. A
// wave owns a strip of 62 columns (lanes 1..62; lanes 0 and 63 only carry the
// neighbouring columns) and walks kDetRows rows of it.
// Per row it loads one value per pyramid level and lane (six coalesced 256-byte
// loads, issued a group of rows ahead), forms the five DoG values (level l+1 -
// level l, kDiff of sift.cpp:256-260), obtains the left / right neighbours
// from the adjacent lanes, and keeps the 3x3 neighbourhood of every image in
// registers, sliding down one row at a time. The test is branch-free:
// border, |D| > 0.0108333, and "v > every neighbour" evaluated as
// v > max(neighbours) -- maxP leaves out (y-1, x+1), the position the
// reference's test of the previous image never looks at (sift.cpp:299-301),
// maxC leaves out the centre, max9 takes all nine; likewise for minima.
// The survivors set their bit in the candidate bitmask (cleared beforehand)
// with an integer atomic (order independent). The refinement, which needs many
// more registers than this loop, is a kernel of its own (k_refine).
constexpr int kDetCols = 62;        // output columns per wave
// (double)|v| > 0.0108333 (sift.cpp:288, the literal is a double) holds for a float v exactly when
// |v| exceeds the largest float that is not above the literal, so the test needs no conversion.
constexpr float kContrPre = ((double)(float)K_CONTR_THR_1 <= K_CONTR_THR_1)
                                ? (float)K_CONTR_THR_1
                                : __builtin_nextafterf((float)K_CONTR_THR_1, 0.0f);
constexpr int kDetRows = 18;        // rows per block (12 .. 24 measure alike, 32 and more 2.5 % slower)
constexpr int kDetGroup = 3;        // rows loaded ahead

template <bool GRAD>
__global__ __launch_bounds__(256) void k_detect(const float* __restrict__ pyr, size_t pyr_stride,
                                                unsigned long long* __restrict__ cmask, int w, int h, int words,
                                                int nbx, int nby, int nblocks, float2* __restrict__ grad,
                                                size_t grad_stride)
{
    const int item = xcd_contiguous_item(nblocks);
    if (item < 0) return;                        // uniform over the block
    const int f = item / (nbx * nby);
    const int by = (item - f * nbx * nby) / nbx;
    const int bx = item - (f * nby + by) * nbx;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int x = (bx * 4 + wave) * kDetCols + lane - 1;
    const int y0 = by * kDetRows;
    const size_t items = (size_t)w * h;
    pyr += (size_t)f * pyr_stride;
    cmask += (size_t)f * kIntervals * h * words;
    if (GRAD) grad += (size_t)f * grad_stride;
    const bool xgrad = lane >= 1 && lane <= kDetCols && x >= 1 && x <= w - 2;
    const bool xtest = lane >= 1 && lane <= kDetCols && x > kBorder && x < w - kBorder;
    // one row of the six pyramid levels; the five DoG values are formed when the row enters the window
    // Positions outside the image are read from the nearest pixel inside instead: such values only
    // reach pixels closer than kBorder to the edge (extremum test) or on the edge itself (gradients),
    // which are never tested resp. written, so the loads need neither a branch nor a select.
    const int xc = min(max(x, 0), w - 1);
    auto load_row = [&](int y, float v[kLevels]) {
        const float* p = pyr + (size_t)min(max(y, 0), h - 1) * w + xc;
#pragma unroll
        for (int l = 0; l < kLevels; l++) v[l] = p[l * items];
    };
    // nb[l][r][c]: rows r = 0..2 <-> y-1, y, y+1; columns c = 0..2 <-> x-1, x, x+1
    float nb[kLevels - 1][3][3];
    // GRAD: levels 1..3 of this lane's column at rows y-1, y, y+1
    float raw[kIntervals][3];
    auto push_row = [&](const float lv[kLevels]) {
        if (GRAD) {
#pragma unroll
            for (int l = 0; l < kIntervals; l++) {
                raw[l][0] = raw[l][1];
                raw[l][1] = raw[l][2];
                raw[l][2] = lv[l + 1];
            }
        }
        float v[kLevels - 1];
#pragma unroll
        for (int l = 0; l < kLevels - 1; l++) v[l] = lv[l + 1] - lv[l];      // kDiff, sift.cpp:256-260
#pragma unroll
        for (int l = 0; l < kLevels - 1; l++) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                nb[l][0][c] = nb[l][1][c];
                nb[l][1][c] = nb[l][2][c];
            }
            // neighbouring lanes through DPP whole-wave shifts (no LDS traffic): wave_shr:1
            // delivers lane i-1's value to lane i, wave_shl:1 lane i+1's; the end lanes,
            // which receive nothing, are the two halo lanes whose results are not used
            const int vi = __builtin_bit_cast(int, v[l]);
            nb[l][2][0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x138, 0xf, 0xf, false));
            nb[l][2][1] = v[l];
            nb[l][2][2] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x130, 0xf, 0xf, false));
        }
    };
    {
        float v[kLevels];
#pragma unroll
        for (int l = 0; l < kLevels - 1; l++)
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) nb[l][r][c] = 0.0f;
#pragma unroll
        for (int l = 0; l < kIntervals; l++)
#pragma unroll
            for (int r = 0; r < 3; r++) raw[l][r] = 0.0f;
        load_row(y0 - 1, v);
        push_row(v);
        load_row(y0, v);
        push_row(v);
    }
    for (int g0 = 0; g0 < kDetRows && y0 + g0 < h; g0 += kDetGroup) {      // uniform over the block
        float ahead[kDetGroup][kLevels];
#pragma unroll
        for (int g = 0; g < kDetGroup; g++) load_row(y0 + g0 + g + 1, ahead[g]);
#pragma unroll
        for (int g = 0; g < kDetGroup; g++) {
            const int y = y0 + g0 + g;
            push_row(ahead[g]);                             // window now centred on row y
            if (g0 + g >= kDetRows) continue;               // the last group may reach past the block's rows
            if (GRAD) {
                // gradient planes of levels 1..3 at (x, y), see k_gradient: the same operations on the
                // same operands, so the same bits
                const bool gok = xgrad && y >= 1 && y <= h - 2;
#pragma unroll
                for (int l = 0; l < kIntervals; l++) {
                    const int ci = __builtin_bit_cast(int, raw[l][1]);
                    const float lf = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ci, 0x138, 0xf, 0xf, false));
                    const float rt = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ci, 0x130, 0xf, 0xf, false));
                    const float dx = rt - lf;
                    const float dy = raw[l][0] - raw[l][2];
                    const float mag = __builtin_sqrtf(dx * dx + dy * dy);
                    const float theta = mif_atan2f(dy, dx);
                    const uint32_t mb = (mif_float_to_bits(mag) & 0x7fffffffu) | (dy != 0.0f ? 0x80000000u : 0u);
                    const v2f val = (v2f){mif_bits_to_float(mb), theta};
                    if (gok) *(v2f*)&grad[(size_t)l * items + (size_t)y * w + x] = val;
                }
            }
            float maxP[3], minP[3], maxC[3], minC[3], max9[3], min9[3], cen[3];
#pragma unroll
            for (int l = 0; l < kLevels - 1; l++) {
                const float (*v)[3] = nb[l];
                const float rmx0 = fmaxf(fmaxf(v[0][0], v[0][1]), v[0][2]), rmn0 = fminf(fminf(v[0][0], v[0][1]), v[0][2]);
                const float rmx1 = fmaxf(fmaxf(v[1][0], v[1][1]), v[1][2]), rmn1 = fminf(fminf(v[1][0], v[1][1]), v[1][2]);
                const float rmx2 = fmaxf(fmaxf(v[2][0], v[2][1]), v[2][2]), rmn2 = fminf(fminf(v[2][0], v[2][1]), v[2][2]);
                if (l <= 2) {           // previous image of level l + 1
                    maxP[l] = fmaxf(fmaxf(fmaxf(v[0][0], v[0][1]), rmx1), rmx2);
                    minP[l] = fminf(fminf(fminf(v[0][0], v[0][1]), rmn1), rmn2);
                }
                if (l >= 1 && l <= 3) { // current image of level l
                    maxC[l - 1] = fmaxf(fmaxf(rmx0, fmaxf(v[1][0], v[1][2])), rmx2);
                    minC[l - 1] = fminf(fminf(rmn0, fminf(v[1][0], v[1][2])), rmn2);
                    cen[l - 1] = v[1][1];
                }
                if (l >= 2) {           // next image of level l - 1
                    max9[l - 2] = fmaxf(fmaxf(rmx0, rmx1), rmx2);
                    min9[l - 2] = fminf(fminf(rmn0, rmn1), rmn2);
                }
            }
            const bool inside = xtest && y > kBorder && y < h - kBorder;
#pragma unroll
            for (int s = 1; s <= kIntervals; s++) {
                const float cur = cen[s - 1];
                const bool cand = inside && mif_fabsf(cur) > kContrPre &&
                                  ((cur > maxP[s - 1] && cur > maxC[s - 1] && cur > max9[s - 1]) ||
                                   (cur < minP[s - 1] && cur < minC[s - 1] && cur < min9[s - 1]));
                if (cand)                                               // about one lane in a hundred
                    atomicOr(&cmask[((size_t)(s - 1) * h + y) * words + (x >> 6)], 1ull << (x & 63));
            }
        }
    }
}

// Refinement of the candidates (sift.cpp:323-417): one thread per 64-bit word of the candidate
// bitmask. The thread refines every candidate of its word, writes the word of the acceptance
// bitmask (all words, so that mask needs no clearing) and adds the accepted ones to the row total.
__global__ __launch_bounds__(256) void k_refine(const float* __restrict__ pyr, size_t pyr_stride,
                                                const unsigned long long* __restrict__ cmask,
                                                unsigned long long* __restrict__ mask, int* __restrict__ rowcnt,
                                                int w, int h, int words, int o, int nwords_total)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;    // word index over (frame, level, row, word)
    if (i >= nwords_total) return;
    unsigned long long c = cmask[i];
    unsigned long long acc = 0ull;
    if (c != 0ull) {
        const int r_all = i / words;                        // (frame * 3 + level) * h + y
        const int k = i - r_all * words;
        const int fl = r_all / h;
        const int y = r_all - fl * h;
        const int f = fl / kIntervals;
        const int s = fl - f * kIntervals + 1;
        const float* P = pyr + (size_t)f * pyr_stride;
        const size_t items = (size_t)w * h;
        while (c != 0ull) {
            const int bit = __ffsll((long long)c) - 1;
            c &= c - 1ull;
            float4 kp;
            if (refine_pixel(P, items, w, h, s, o, k * 64 + bit, y, &kp)) acc |= 1ull << bit;
        }
        if (acc != 0ull) atomicAdd(&rowcnt[r_all], __popcll(acc));
    }
    mask[i] = acc;
}

// Exclusive scan of a frame's per-row counts (rows ordered level-major, then
// y), the octave's keypoint count (cut at max_features = items/100,
// sift.cpp:1356, 408) and its raw total. One block of 1024 threads per frame.
__global__ __launch_bounds__(1024) void k_scan_rows(const int* __restrict__ rowcnt, int* __restrict__ rowoff,
                                                    int rows, int maxf, int o, int* __restrict__ counts)
{
    __shared__ int s_part[1024];
    rowcnt += (size_t)blockIdx.x * rows;
    rowoff += (size_t)blockIdx.x * rows;
    counts += (size_t)blockIdx.x * CNT_WORDS;
    const int tid = threadIdx.x;
    const int per = (rows + 1023) / 1024;
    const int lo = tid * per;
    const int hi = min(rows, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; i++) sum += rowcnt[i];
    s_part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = (tid >= d) ? s_part[tid - d] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - sum;
    for (int i = lo; i < hi; i++) {
        rowoff[i] = run;
        run += rowcnt[i];
    }
    if (tid == 1023) {
        const int total = s_part[1023];
        const int n = min(total, maxf);
        counts[CNT_OCT0 + o] = n;
        counts[CNT_RAW0 + o] = total;
        int cum = n;
        for (int i = 0; i < o; i++) cum += counts[CNT_OCT0 + i];
        counts[CNT_TOTAL] = cum;
    }
}

__device__ __forceinline__ int octave_base(const int* counts, int o)
{
    int b = 0;
    for (int i = 0; i < o; i++) b += counts[CNT_OCT0 + i];
    return b;
}

// Emission pass: one thread per keypoint slot p of (frame, octave). The thread
// finds the row (level, y) whose range [rowoff, rowoff + rowcnt) holds p by
// bisection over the exclusive scan, then the (p - rowoff)-th set bit of that
// row's bitmask, re-refines the pixel (the refinement is a pure function of the
// DoG stack) and writes the keypoint at base(octave) + p, i.e. in raster order.
__global__ __launch_bounds__(256) void k_emit(const float* __restrict__ pyr, size_t pyr_stride,
                                              const unsigned long long* __restrict__ mask,
                                              const int* __restrict__ rowcnt, const int* __restrict__ rowoff, int w,
                                              int h, int words, int o, int maxf, int cap,
                                              const int* __restrict__ counts, float4* __restrict__ kp)
{
    const int f = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    counts += (size_t)f * CNT_WORDS;
    if (p >= counts[CNT_OCT0 + o]) return;              // count is already cut at maxf
    const int rows = kIntervals * h;
    rowcnt += (size_t)f * rows;
    rowoff += (size_t)f * rows;
    // last row with rowoff <= p; rows without keypoints share their offset with the next
    // row, so the last one among equals is the row that owns p
    int lo = 0, hi = rows - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (rowoff[mid] <= p) lo = mid; else hi = mid - 1;
    }
    const int r = lo;
    int j = p - rowoff[r];                              // rank of the pixel within its row
    const unsigned long long* mrow = mask + ((size_t)f * rows + r) * words;
    int x = -1;
    for (int k = 0; k < words; k++) {
        unsigned long long m = mrow[k];
        const int c = __popcll(m);
        if (j < c) {
            for (; j > 0; j--) m &= m - 1ull;           // drop the lower set bits
            x = k * 64 + __ffsll((long long)m) - 1;
            break;
        }
        j -= c;
    }
    if (x < 0) return;                                  // cannot happen: rowcnt counts the set bits
    const int s = r / h + 1;
    const int y = r - (s - 1) * h;
    const size_t items = (size_t)w * h;
    pyr += (size_t)f * pyr_stride;
    float4 out;
    refine_pixel(pyr, items, w, h, s, o, x, y, &out);
    kp[(size_t)f * cap + octave_base(counts, o) + p] = out;
}

// ------------------------------------------------------------------ gradients

// Gradient magnitude and angle of pyramid levels 1..3, once per pixel. The
// orientation and descriptor kernels of the reference recompute
//   dx = L(y, x+1) - L(y, x-1),  dy = L(y-1, x) - L(y+1, x)   (sift.cpp:483-487; the
//   descriptor uses dy with the opposite sign, sift.cpp:665-669)
//   mag = sqrt(dx*dx + dy*dy),  theta = atan2(dy, dx)
// for every sample of every keypoint window; windows overlap heavily, so the
// values are computed here once and gathered there. Stored per pixel as
// float2 {mag with the sign bit set iff dy != 0, theta of the orientation
// kernel's convention}; the descriptor's angle atan2(-dy, dx) is -theta when
// dy != 0 and theta when dy == 0 (mif_atan2f is odd in y except at y = 0), so
// both kernels read back exactly the values they would have computed.
// Border pixels are never sampled (windows are clipped to [1, w-2] x [1, h-2])
// and are not written. A thread owns one column of eight consecutive rows and
// issues all 26 loads before the first use, which keeps enough bytes in flight
// per wave to cover the HBM latency. grid (ceil(w/128), ceil(h/16), 3 levels x frames).
constexpr int kGradRows = 8;

__global__ __launch_bounds__(256) void k_gradient(const float* __restrict__ pyr, size_t pyr_stride,
                                                  float2* __restrict__ grad, size_t grad_stride, int w, int h)
{
    // a block covers 128 columns x 2 groups of rows (two waves each): the rows the two groups share
    // are fetched by one CU at about the same time and meet in its L1 (256 x 1: +8 % time, 64 x 4: +3 %)
    const int x = blockIdx.x * 128 + (threadIdx.x & 127);
    const int y0 = (blockIdx.y * 2 + (threadIdx.x >> 7)) * kGradRows;
    const int f = blockIdx.z / kIntervals;
    const int l = blockIdx.z - f * kIntervals;            // 0..2 -> pyramid level l + 1
    if (x < 1 || x > w - 2) return;
    const size_t items = (size_t)w * h;
    const float* img = pyr + (size_t)f * pyr_stride + (size_t)(l + 1) * items;
    float2* out = grad + (size_t)f * grad_stride + (size_t)l * items;
    float c[kGradRows + 2], lf[kGradRows], rt[kGradRows];
#pragma unroll
    for (int r = 0; r < kGradRows + 2; r++) {
        const int y = min(max(y0 + r - 1, 0), h - 1);
        c[r] = img[y * w + x];
    }
#pragma unroll
    for (int r = 0; r < kGradRows; r++) {
        const int y = min(y0 + r, h - 1);
        lf[r] = img[y * w + x - 1];
        rt[r] = img[y * w + x + 1];
    }
#pragma unroll
    for (int r = 0; r < kGradRows; r++) {
        const int y = y0 + r;
        if (y < 1 || y > h - 2) continue;
        const float dx = rt[r] - lf[r];
        const float dy = c[r] - c[r + 2];
        const float mag = __builtin_sqrtf(dx * dx + dy * dy);
        const float theta = mif_atan2f(dy, dx);
        const uint32_t mb = (mif_float_to_bits(mag) & 0x7fffffffu) | (dy != 0.0f ? 0x80000000u : 0u);
        // streamed past the L2 (nothing reads the planes before the launch ends), which keeps the
        // rows that neighbouring blocks share resident there
        const v2f val = (v2f){mif_bits_to_float(mb), theta};
        __builtin_nontemporal_store(val, (v2f*)&out[(size_t)y * w + x]);
    }
}

// --------------------------------------------------------------- orientation

// sift.cpp:431-582 kOrientation. A block (one wave) serves four keypoints at a
// time, 16 lanes each, exactly the reference's work-group of 16: lane l
// accumulates window pixels l, l+16, ... into its private 36-bin histogram and
// the 16 histograms are summed with the reference's tree (8, 4, 2, 1). The two
// smoothing passes and the search for the last qualifying peak, which the
// reference leaves to work-item 0, are shared by the 16 lanes: every bin of a
// pass depends on the previous pass only, so the values are the same.
__global__ __launch_bounds__(64) void k_orientation(const float4* __restrict__ kp, const float2* __restrict__ grad,
                                                    size_t grad_stride, int w, int h, int o, int cap,
                                                    const int* __restrict__ counts, float* __restrict__ orient,
                                                    OriBins bins)
{
    __shared__ float s_hist[4][kOriLanes][kOriBins + 1];
    __shared__ float s_thr[kOriBins + 2];
    if (threadIdx.x < kOriBins + 2) s_thr[threadIdx.x] = bins.t[threadIdx.x];
    const int f = blockIdx.y;
    counts += (size_t)f * CNT_WORDS;
    kp += (size_t)f * cap;
    orient += (size_t)f * cap;
    grad += (size_t)f * grad_stride;
    const int grp = threadIdx.x >> 4;
    const int lid = threadIdx.x & 15;
    const int n = counts[CNT_OCT0 + o];
    const int first = octave_base(counts, o);
    float* hist = s_hist[grp][lid];
    // a block (one wave) takes four keypoints at a time and walks the frame's list with
    // the grid's stride; the list capacity is far larger than the list
    for (int q = blockIdx.x; q * 4 < n; q += gridDim.x) {      // uniform over the block
        const int g = q * 4 + grp;
        const bool active = g < n;
        const int gi = first + g;
        for (int i = 0; i < kOriBins; i++) hist[i] = 0.0f;
        __syncthreads();                                    // s_thr is complete; the last round's reads are done

        if (active) {
            const float4 f4 = kp[gi];
            const float exp_denom = (float)((double)f4.z * K_ORI_SIG_FCTR);
            const int rad = (int)mif_round(K_ORI_RADIUS * (double)f4.z);
            const float exp_factor = -0.5f / (exp_denom * exp_denom);
            const int sq_thres = rad * rad;
            // the reference offsets the pyramid by (int)(w*h*level) floats (sift.cpp:459); the
            // gradient planes are stored from level 1 on
            const int offset_py = (int)((float)(w * h) * f4.w) - w * h;
            const float p2o = mif_pow2i(o);
            const int x = (int)(f4.x / p2o);
            const int y = (int)(f4.y / p2o);
            const int xmin = max(1, x - rad), ymin = max(1, y - rad);
            const int xmax = min(w - 2, x + rad), ymax = min(h - 2, y + rad);
            const int wx = xmax - xmin + 1, hy = ymax - ymin + 1;
            const int loops = wx * hy;
            const float2* img = grad + offset_py;
            // window pixel i = (i / wx + ymin, i % wx + xmin), i = lid, lid + 16, ...
            // the window is clipped to [1, w-2] x [1, h-2], so every window pixel can be gathered.
            // Several pixels per turn (i, i + 16, ...): a pixel costs the lane three LDS round trips in a row
            // (two bin thresholds, then the histogram entry), and with 9.5 KB of histograms per wave only
            // four waves share a SIMD, so those latencies are the kernel's time. Both thresholds of a
            // pixel come with one read, the turn's reads go out together, and the gathers of the next
            // turn's pixels are in flight meanwhile. A pixel outside the circle (or past the window's
            // end) adds +0 to bin 0, which leaves every value a histogram can hold as it is.
            auto step = [&](int& px, int& py) {
                px += kOriLanes;
                while (px > xmax) { px -= wx; py++; }
            };
            constexpr int U = kOriUnroll;            // window pixels per turn
            int px[U], py[U];
            float2 gn[U];
            px[0] = lid % wx + xmin;
            py[0] = lid / wx + ymin;
#pragma unroll
            for (int u = 1; u < U; u++) {
                px[u] = px[u - 1];
                py[u] = py[u - 1];
                step(px[u], py[u]);
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                gn[u] = (lid + u * kOriLanes < loops) ? img[py[u] * w + px[u]] : make_float2(0.0f, 0.0f);
            for (int i = lid; i < loops; i += U * kOriLanes) {
                int xx[U], yy[U];
                float2 gg[U];
#pragma unroll
                for (int u = 0; u < U; u++) { xx[u] = px[u]; yy[u] = py[u]; gg[u] = gn[u]; }
                px[0] = px[U - 1];
                py[0] = py[U - 1];
                step(px[0], py[0]);
#pragma unroll
                for (int u = 1; u < U; u++) {
                    px[u] = px[u - 1];
                    py[u] = py[u - 1];
                    step(px[u], py[u]);
                }
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (i + (U + u) * kOriLanes < loops) gn[u] = img[py[u] * w + px[u]];
                int bin[U], est[U];
                float wgt[U], tlo[U], thi[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const float ddx = (float)(xx[u] - x), ddy = (float)(yy[u] - y);
                    const int sq_dist = (int)(ddx * ddx + ddy * ddy);
                    ok[u] = i + u * kOriLanes < loops && xx[u] > 0 && xx[u] < w - 1 && yy[u] > 0 && yy[u] < h - 1 &&
                            sq_dist <= sq_thres;
                    // bin = round(36 (theta + pi) / (2 pi)) in double (sift.cpp:493), a monotone step
                    // function of theta: a float estimate that is off by at most one, corrected against
                    // the exact thresholds of the estimate's bin
                    est[u] = min(max((int)__builtin_fmaf(gg[u].y, 5.72957795130823f, 18.5f), 0), kOriBins);
                    tlo[u] = s_thr[est[u]];
                    thi[u] = s_thr[est[u] + 1];
                    // sq_dist <= rad^2 with rad = round(4.5 sigma) and exp_factor = -0.5 / (1.5 sigma)^2: the
                    // argument stays above -6 inside the circle (and above -12 in the window's corners)
                    wgt[u] = mif_fabsf(gg[u].x) * mif_expf_bounded((float)sq_dist * exp_factor);
                }
                float hv[U], res[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const float theta = gg[u].y;
                    int bn = est[u] + ((theta >= thi[u]) ? 1 : 0) - ((theta < tlo[u]) ? 1 : 0);
                    bn = (bn < kOriBins) ? bn : 0;
                    bin[u] = ok[u] ? bn : 0;
                    wgt[u] = ok[u] ? wgt[u] : 0.0f;
                    hv[u] = hist[bin[u]];
                }
                // the additions in window order: a pixel whose bin an earlier pixel of the turn has
                // already met continues from that pixel's result
#pragma unroll
                for (int u = 0; u < U; u++) {
                    float from = hv[u];
#pragma unroll
                    for (int v = 0; v < u; v++) from = (bin[v] == bin[u]) ? res[v] : from;
                    res[u] = from + wgt[u];
                }
#pragma unroll
                for (int u = 0; u < U; u++) hist[bin[u]] = res[u];
            }
        }
        __syncthreads();
        // tree reduction of the 16 private histograms, association as sift.cpp:509-522
        for (int i = lid; i < kOriBins; i += kOriLanes) {
            float v[kOriLanes];
#pragma unroll
            for (int l = 0; l < kOriLanes; l++) v[l] = s_hist[grp][l][i];
#pragma unroll
            for (int l = 0; l < 8; l++) v[l] += v[l + 8];
#pragma unroll
            for (int l = 0; l < 4; l++) v[l] += v[l + 4];
#pragma unroll
            for (int l = 0; l < 2; l++) v[l] += v[l + 2];
            v[0] += v[1];
            s_hist[grp][0][i] = v[0];     // only this lane touches column i
        }
        __syncthreads();

        // Two [1/4 1/2 1/4] passes in double (sift.cpp:531-543). The reference's single work-item
        // updates in place but feeds every bin the OLD values of both neighbours, so the bins of a
        // pass are independent: the 16 lanes share them, rows 0 -> 1 -> 2 of the group's array.
        for (int j = 0; j < 2; j++) {
            const float* in = s_hist[grp][j];
            float* out = s_hist[grp][j + 1];
            for (int i = lid; i < kOriBins; i += kOriLanes) {
                const float hp = in[(i == 0) ? (kOriBins - 1) : (i - 1)];
                const float hc = in[i];
                const float hn = in[(i + 1 == kOriBins) ? 0 : (i + 1)];
                out[i] = (float)(0.25 * (double)hp + 0.5 * (double)hc + 0.25 * (double)hn);
            }
            __syncthreads();
        }
        // The reference scans the bins upwards and keeps the LAST peak that reaches 0.8 of the
        // maximum (sift.cpp:545-580): every lane tests its bins, the highest bin index wins.
        const float* hs = s_hist[grp][2];
        float v = hs[0];
        for (int i = 1; i < kOriBins; i++)
            if (hs[i] > v) v = hs[i];
        v = (float)(K_ORI_PEAK_RATIO * (double)v);
        int best = -1;
        float result = -1.0f;               // _ORIENT_INIT_CONST
        for (int i = lid; i < kOriBins; i += kOriLanes) {
            const float hc = hs[i];
            const float hp = hs[(i == 0) ? (kOriBins - 1) : (i - 1)];
            const float hn = hs[(i + 1) % kOriBins];
            if (hc >= v && hc > hn && hc > hp) {
                float fbin = (float)((double)(float)i +
                                     ((double)(0.5f * (hp - hn)) / ((double)hp - 2.0 * (double)hc + (double)hn)));
                fbin = (fbin < 0.0f) ? (fbin + (float)kOriBins)
                                     : ((fbin >= (float)kOriBins) ? (fbin - (float)kOriBins) : fbin);
                result = (float)(((K_M_PI2 * (double)fbin) / kOriBins) - K_M_PI);
                best = i;
            }
        }
        // highest qualifying bin of the 16 lanes (xor shuffles stay inside the group of 16)
        int top = best;
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) top = max(top, __shfl_xor(top, m, 64));
        if (active && (top < 0 ? lid == 0 : best == top)) orient[gi] = result;
    }
}

// ---------------------------------------------------------------- descriptor

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// The private 9-bin histograms of a block's 128 threads lie bin-major in LDS, s_des[bin][thread]: a
// thread's bins are kDescStride floats apart, so whatever bins the lanes of a wave update, lane t is in
// bank t % 32 and no two lanes of a half-wave meet in a bank. (Thread-major, 9 floats per thread, the
// bank is (9 t + bin) % 32 and lanes with different bins collide: 3.95 conflict cycles per access.)
#ifndef MIF_DESC_STRIDE
#define MIF_DESC_STRIDE 128
#endif
constexpr int kDescStride = MIF_DESC_STRIDE;       // 128: bin-major; 1: thread-major (the form kept for comparison)
constexpr int kDescThread = kDescStride == 1 ? 9 : 1;

// One accepted window sample of kDescriptor (sift.cpp:669-699): weights, orientation bin and the
// update of the lane's private histogram d[9]. (nx, ny) are the sample's coordinates in units of the
// cell size relative to the cell centre, |nx|, |ny| < 1; g2 is the gradient plane's entry.
__device__ __forceinline__ void desc_accumulate(float* __restrict__ d, float nx, float ny, float2 g2, float offx,
                                                float offy, float ang)
{
    const float nxn = mif_fabsf(nx), nyn = mif_fabsf(ny);
    const float mod = mif_fabsf(g2.x);
    // atan2(-dy, dx) from the stored atan2(dy, dx): sign bit of g2.x is set iff dy != 0
    float th = mif_bits_to_float(mif_float_to_bits(g2.y) ^ (mif_float_to_bits(g2.x) & 0x80000000u));
    const float dnx = nx + offx, dny = ny + offy;
    // |dnx|, |dny| < 2.5 here, so the argument lies in (-1.6, 0]
    const float ww = mif_expf_bounded(-0.125f * (dnx * dnx + dny * dny));
    // 1.0 - nxn is evaluated in double in the reference (sift.cpp:675-676); the
    // difference is exact there, so rounding it to float equals the float subtraction
    const float wwx = 1.0f - nxn;
    const float wwy = 1.0f - nyn;
    const float wgt = ww * wwx * wwy * mod;
    th -= ang;
    // The reference wraps with two loops in double, `while (th < 0) th += 2 pi` and
    // `while (th >= 2 pi) th -= 2 pi`. |theta| and |ang| are at most float(pi), so
    // th lies in [-2 float(pi), 2 float(pi)]: the first loop runs at most twice
    // (float(pi) > pi) and the second at most once. (double)th >= 2 pi holds exactly
    // when th >= float(2 pi), the smallest float above 2 pi, so that test needs no
    // conversion; 64-bit operations are slow on this GPU.
    // The first round of the first loop without a branch: th + s 2 pi with s = 1.0 or 0.0 in one fused
    // step (one rounding, like the addition; s = 0.0 leaves th as it is). Whatever else the loops do is
    // rare and sits behind a test of the whole wave.
    th = (float)__builtin_fma(th < 0.0f ? 1.0 : 0.0, K_M_PI2, (double)th);
    // one unsigned comparison of the bits finds both: negative (sign bit) and >= float(2 pi) = 0x40c90fdb
    if (__builtin_amdgcn_uicmp(mif_float_to_bits(th), 0x40c90fdbu, 35) != 0) {     // 35: unsigned >=
        if (th < 0.0f) th = (float)((double)th + K_M_PI2);
        if (th >= 6.2831854820251465f) {
            th = (float)((double)th - K_M_PI2);
            if (th >= 6.2831854820251465f) th = (float)((double)th - K_M_PI2);
        }
    }
    const float tth = (float)((double)th * K_RPI);
    // th is in [0, float(2 pi)) now. The largest such float, 0x40c90fda, gives tth = 7.9999995 (its product
    // with 4 / pi is nearer to that float than to 8), so tth lies in [0, 8), the bin (int)tth in 0..7 and the
    // reference's `% 8` changes nothing; the truncation is the floor.
    // (the fraction instruction returns tth - floor(tth), an exact difference for tth >= 0)
    const int fo = (int)tth;
    const float do0 = __builtin_amdgcn_fractf(tth);
    d[fo * kDescStride] += (1.0f - do0) * wgt;
    d[(fo + 1) * kDescStride] += do0 * wgt;
}

// sift.cpp:585-709 kDescriptor fused with sift.cpp:712-765 kNormalize. One
// block of 128 threads per keypoint: 16 cells x 8 lanes, lane l of a cell
// accumulates window pixels l, l+8, ... into a private 9-bin histogram; bin 8
// folds into bin 0 and the 8 histograms of a cell are summed in lane order.
// Normalisation sums the 128 squares sequentially, as the reference's
// one-thread-per-keypoint kernel does.
// MODE 0: every window goes through the general loop (the form kept for comparison). MODE 1: waves whose
// windows are all at least 8 pixels wide take the streamlined loop below. MODE 2: such waves whose windows
// are also at most 128 x 128 pixels first mark the pixels that pass the cell test (a loop without gathers
// or histogram work) and then visit only those.
template <int MODE>
__global__ __launch_bounds__(128) void k_descriptor(const float4* __restrict__ kp, const float* __restrict__ orient,
                                                    const float2* __restrict__ grad, size_t grad_stride, int w, int h,
                                                    int o, int cap, const int* __restrict__ counts,
                                                    float* __restrict__ desc)
{
    __shared__ float s_des[9 * 128];
    __shared__ float s_val[128];
    __shared__ float s_sq[128];
    __shared__ float s_norm;
    const int f = blockIdx.y;
    counts += (size_t)f * CNT_WORDS;
    kp += (size_t)f * cap;
    orient += (size_t)f * cap;
    desc += (size_t)f * cap * 128;
    grad += (size_t)f * grad_stride;
    const int n = counts[CNT_OCT0 + o];
    const int first = octave_base(counts, o);
    const int tid = threadIdx.x;
    const int lid = tid & 7;
    const int bid = tid >> 3;
    const int ix = bid & 3, iy = bid >> 2;
    float* d = s_des + tid * kDescThread;
    // the grid's x extent is far smaller than the list capacity (most of which stays empty):
    // a block walks the frame's keypoints with the grid's stride
    for (int k = blockIdx.x; k < n; k += gridDim.x) {      // uniform over the block
    const int gi = first + k;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i * kDescStride] = 0.0f;

    const float4 f4 = kp[gi];
    const float p2o = mif_pow2i(o);
    const int x = (int)(f4.x / p2o);
    const int y = (int)(f4.y / p2o);
    const float2* img = grad + ((int)((float)(w * h) * f4.w) - w * h);    // planes start at level 1
    const float ang = -orient[gi];
    const float spt = (float)fabs((double)f4.z * 3.0);
    float sn, cs;
    mif_sincosf(ang, &sn, &cs);
    const float cspt = cs * spt, sspt = sn * spt;
    const float crspt = cs / spt, srspt = sn / spt;
    const float offx = (float)ix - 1.5f, offy = (float)iy - 1.5f;
    const float ptx = cspt * offx - sspt * offy + (float)x;
    const float pty = cspt * offy + sspt * offx + (float)y;
    const float bsz = mif_fabsf(cspt) + mif_fabsf(sspt);
    // MAXV / MINV compare an int with a float in float and truncate back (sift.cpp:641-644)
    const float fxmin = __builtin_floorf(ptx - bsz), fymin = __builtin_floorf(pty - bsz);
    const float fxmax = __builtin_floorf(ptx + bsz), fymax = __builtin_floorf(pty + bsz);
    const int xmin = (1.0f >= fxmin) ? 1 : (int)fxmin;
    const int ymin = (1.0f >= fymin) ? 1 : (int)fymin;
    const int xmax = ((float)(w - 2) <= fxmax) ? (w - 2) : (int)fxmax;
    const int ymax = ((float)(h - 2) <= fymax) ? (h - 2) : (int)fymax;
    const int wx = xmax - xmin + 1, hy = ymax - ymin + 1;
    // a window that is empty in both directions has a positive wx*hy in the
    // reference and would index outside the image; it is treated as empty
    const int loops = (wx > 0 && hy > 0) ? wx * hy : 0;
    // Streamlined walk for waves whose windows are all at least 8 pixels wide (windows cut by the image
    // border can be narrower): a step of 8 pixels then wraps into the next row at most once, so the
    // walk is a handful of selects; coordinates run as floats (small integers: the additions are exact),
    // a lane is inside its window exactly while its row is, and the gathers go through a buffer
    // descriptor of the level's plane, whose range check makes the look-ahead of lanes that have left
    // their window harmless (they read 0 and accumulate nothing). The plane must fit a 32-bit offset.
    const bool wide = MODE != 0 && __all(wx >= 8 && hy > 0) && (size_t)w * h < ((size_t)1 << 29);
    if (MODE == 2 && wide && __all(wx <= 128 && hy <= 128) && w <= (1 << 14) && (size_t)w * h <= ((size_t)1 << 24)) {
        // Two phases per 32 rounds of the walk. Phase 1 runs the walk and the cell test alone and leaves
        // bit j of `mask` set when the lane's pixel of round j passes. Phase 2 visits the set bits in
        // ascending order (the order in which the reference's work-item adds to its histogram): window
        // pixel i = lid + 8 (base + j) lies in row q = i / wx and column i % wx. The visit carries
        // ix = i + xmin (an integer below 2^15, exact in float) and finds q as
        // floor(ix / wx + (0.5 - xmin) / wx) in float: the exact value (i + 0.5) / wx is at least
        // 0.5 / 128 = 3.9e-3 away from an integer (wx <= 128), and the float evaluation is off by less than
        // 7e-4 (1 / wx and the constant rounded once each, values of at most 2^15 / 8, one rounding of
        // the sum), so the floor is exact. Column and row then are x = ix - q wx and y = q + ymin, and
        // the pixel's index y w + x is an integer below 2^24: all exact in float.
        // About 40 % of a window's pixels fail the test and cost the first phase only.
        const uint32_t lvl = __builtin_amdgcn_readfirstlane((uint32_t)((int)((float)(w * h) * f4.w) - w * h));
        const float2* plane = grad + lvl;
        const uint64_t pb = (uint64_t)plane;
        const uint32_t pb_lo = __builtin_amdgcn_readfirstlane((uint32_t)pb);
        const uint32_t pb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32));
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)pb_hi << 32) | pb_lo), 0,
                                                            (int)((uint32_t)(w * h) * 8u), 0x00020000);
        const float fxmax = (float)xmax, fymax = (float)ymax, fwx = (float)wx, fw = (float)w;
        const float fxmin = (float)xmin, fymin = (float)ymin;
        const float rwx = 1.0f / fwx, qoff = 0.5f * rwx - fxmin * rwx;
        float fx = (float)(xmin + lid), fy = (float)ymin;
        float fbase = (float)(xmin + lid);                  // xmin + window pixel index of this lane in round `base`
        // a lane's rounds: ceil((loops - lid) / 8); the wave runs to its longest lane
        int most = (loops - lid + 7) >> 3;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) most = max(most, __shfl_xor(most, m, 64));
        most = __builtin_amdgcn_readfirstlane(most);
        for (int base = 0; base < most; base += 32, fbase += 256.0f) {
            const int nj = __builtin_amdgcn_readfirstlane(min(32, most - base));
            // round j's bit enters at the bottom and moves up: reversed and shifted down afterwards
            uint32_t rmask = 0;
            for (int j = 0; j < nj; j++) {
                const float dx = fx - ptx;
                const float dy = fy - pty;
                const float nx = crspt * dx + srspt * dy;
                const float ny = crspt * dy - srspt * dx;
                // rmask = 2 rmask + pass in one instruction: the lanes' `pass` bits are the carry-in
                const uint64_t passed = __builtin_amdgcn_ballot_w64(fy <= fymax) &
                                        __builtin_amdgcn_ballot_w64(mif_fabsf(nx) < 1.0f) &
                                        __builtin_amdgcn_ballot_w64(mif_fabsf(ny) < 1.0f);
                asm("v_addc_co_u32 %0, vcc, %1, %1, %2" : "=v"(rmask) : "v"(rmask), "s"(passed) : "vcc");
                fx += 8.0f;
                const bool wrap = fx > fxmax;
                fx -= wrap ? fwx : 0.0f;
                fy += wrap ? 1.0f : 0.0f;
            }
            uint32_t mask = __builtin_bitreverse32(rmask) >> (32 - nj);      // nj is 1..32
            // phase 2, the gather of the next marked pixel in flight while the current one is accumulated
            // (two picks used alternately: nothing is copied from "next" to "current")
            struct Pick {
                float cx, cy;
                v2f g;
                bool act;
                uint64_t any;       // lanes of the wave with a pick (a scalar: the loop's exit test)
            };
            auto decode = [&](Pick& p) {
                p.act = mask != 0;
                p.any = __builtin_amdgcn_uicmp(mask, 0u, 33);      // 33: "not equal"; the compare's lane mask itself
                // lowest set bit; -1 from an empty mask (the instruction defines it), whose lane computes
                // a position it never uses
                int j;
                asm("v_ffbl_b32 %0, %1" : "=v"(j) : "v"(mask));
                mask &= mask - 1;
                const float ix = __builtin_fmaf((float)j, 8.0f, fbase);
                const float q = __builtin_floorf(__builtin_fmaf(ix, rwx, qoff));
                p.cx = __builtin_fmaf(-q, fwx, ix);
                p.cy = q + fymin;
                const uint32_t boff = (uint32_t)__builtin_fmaf(p.cy, fw, p.cx) * 8u;
                // lanes without a marked pixel left read beyond the plane: the range check returns 0
                p.g = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsrc, p.act ? boff : 0xfffffff8u, 0, 0));
            };
            auto visit = [&](const Pick& p) {
                const float dx = p.cx - ptx;
                const float dy = p.cy - pty;
                const float nx = crspt * dx + srspt * dy;
                const float ny = crspt * dy - srspt * dx;
                if (p.act) desc_accumulate(d, nx, ny, make_float2(p.g.x, p.g.y), offx, offy, ang);
            };
            Pick pa, pb;
            decode(pa);
            for (;;) {
                if (pa.any == 0) break;
                decode(pb);
                visit(pa);
                if (pb.any == 0) break;
                decode(pa);
                visit(pb);
            }
        }
    } else if (wide) {
        const uint32_t lvl = __builtin_amdgcn_readfirstlane((uint32_t)((int)((float)(w * h) * f4.w) - w * h));
        const float2* plane = grad + lvl;
        const uint64_t pb = (uint64_t)plane;
        const uint32_t pb_lo = __builtin_amdgcn_readfirstlane((uint32_t)pb);
        const uint32_t pb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32));
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)pb_hi << 32) | pb_lo), 0,
                                                            (int)((uint32_t)(w * h) * 8u), 0x00020000);
        const float fxmax = (float)xmax, fymax = (float)ymax, fwx = (float)wx;
        float fx = (float)(xmin + lid), fy = (float)ymin;
        uint32_t boff = (uint32_t)(ymin * w + xmin + lid) * 8u;
        const uint32_t inc_wrap = (uint32_t)(8 + w - wx) * 8u;
        v2f gn = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsrc, boff, 0, 0));
        while (__any(fy <= fymax)) {
            const float cfx = fx, cfy = fy;
            const v2f g = gn;
            fx += 8.0f;
            const bool wrap = fx > fxmax;
            fx -= wrap ? fwx : 0.0f;
            fy += wrap ? 1.0f : 0.0f;
            boff += wrap ? inc_wrap : 64u;
            gn = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsrc, boff, 0, 0));
            const float dx = cfx - ptx;
            const float dy = cfy - pty;
            const float nx = crspt * dx + srspt * dy;
            const float ny = crspt * dy - srspt * dx;
            // one combined condition (bit-and): a single branch instead of three nested ones
            if ((int)(cfy <= fymax) & (int)(mif_fabsf(nx) < 1.0f) & (int)(mif_fabsf(ny) < 1.0f))
                desc_accumulate(d, nx, ny, make_float2(g.x, g.y), offx, offy, ang);
        }
    } else if (loops > 0) {
        // window pixel i = (i / wx + ymin, i % wx + xmin), i = lid, lid + 8, ...
        // The gather of the next window pixel is issued before the current one is
        // processed (whether or not it will pass the cell test), so its latency overlaps
        // the arithmetic.
        // The coordinates also run as floats (small integers: adding 8 or 1 is exact) and the
        // pixel as a running pointer, which saves the conversions and the index arithmetic.
        int xn = lid % wx + xmin;
        const int y0n = lid / wx + ymin;
        float fxn = (float)xn, fyn = (float)y0n;
        const float fwx = (float)wx;
        const float2* pn = img + (y0n * w + xn);
        const int row_step = w - wx;            // pointer correction when a row of the window ends
        float2 gnext = (lid < loops) ? *pn : make_float2(0.0f, 0.0f);
        for (int i = lid; i < loops; i += 8) {
            const float fx = fxn, fy = fyn;
            const float2 g2 = gnext;
            xn += 8; fxn += 8.0f; pn += 8;
            while (xn > xmax) { xn -= wx; fxn -= fwx; fyn += 1.0f; pn += row_step; }
            if (i + 8 < loops) gnext = *pn;
            float dx = fx - ptx;
            float dy = fy - pty;
            const float nx = crspt * dx + srspt * dy;
            const float ny = crspt * dy - srspt * dx;
            if (mif_fabsf(nx) < 1.0f && mif_fabsf(ny) < 1.0f) desc_accumulate(d, nx, ny, g2, offx, offy, ang);
        }
    }
    d[0] += d[8 * kDescStride];
    __syncthreads();
    float bin = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) bin += s_des[(bid * 8 + i) * kDescThread + lid * kDescStride];

    // normalise -> clamp 0.2 -> normalise -> x512 -> clamp 255
    s_val[tid] = bin;
    s_sq[tid] = bin * bin;
    __syncthreads();
    if (tid == 0) {
        float nrm = 0.0f;
        for (int i = 0; i < 128; i++) nrm += s_sq[i];
        s_norm = 1.0f / __builtin_sqrtf(nrm);
    }
    __syncthreads();
    float v0 = clampf(s_val[tid] * s_norm, 0.0f, (float)K_DESCR_MAG_THR);
    s_sq[tid] = v0 * v0;
    __syncthreads();
    if (tid == 0) {
        float nrm = 0.0f;
        for (int i = 0; i < 128; i++) nrm += s_sq[i];
        s_norm = 1.0f / __builtin_sqrtf(nrm);
    }
    __syncthreads();
    const float v1 = (float)K_INT_DESCR_FCTR * (v0 * s_norm);
    desc[(size_t)gi * 128 + tid] = clampf(v1, 0.0f, 255.0f);
    // the next keypoint's first writes to s_val / s_sq / s_norm come after its first barrier
    }
}

// After the last octave, one thread per frame: status = 0, or -1 when an
// octave found nothing (sift.cpp:1257-1258); in that case the frame's counts
// are cleared so that no stage downstream consumes a partial frame.
__global__ void k_finish(int* counts, int nframes, int want_desc)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nframes) return;
    counts += (size_t)f * CNT_WORDS;
    int status = 0;
    if (want_desc)
        for (int o = 0; o < kOctaves; o++)
            if (counts[CNT_RAW0 + o] < 1) status = MIF_ERR_NO_KEYPOINTS;
    counts[CNT_STATUS] = status;
    if (status != 0) {
        for (int o = 0; o < kOctaves; o++) counts[CNT_OCT0 + o] = 0;
        counts[CNT_TOTAL] = 0;
    }
}

// Readback of a whole batch (sift.cpp:1889-1911 appends features(), orientations() and descriptors()
// octave by octave; a frame's rows are already packed like that at row frame * cap). First row of every
// frame in the packed arrays: exclusive scan of the frames' totals, cut at max_rows.
// offsets[f] = first row of frame f, offsets[nframes] = rows packed, offsets[nframes + 1] = rows the
// batch holds. One block of 1024 threads (a batch has at most 1024 frames).
__global__ __launch_bounds__(1024) void k_pack_offsets(const int* __restrict__ counts, int nframes, int max_rows,
                                                       int* __restrict__ offsets)
{
    __shared__ int s_scan[1024];
    const int f = threadIdx.x;
    const int n = f < nframes ? counts[(size_t)f * CNT_WORDS + CNT_TOTAL] : 0;
    s_scan[f] = n;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int add = f >= d ? s_scan[f - d] : 0;
        __syncthreads();
        s_scan[f] += add;
        __syncthreads();
    }
    const int incl = s_scan[f];
    if (f < nframes) offsets[f] = min(incl - n, max_rows);
    if (f == nframes - 1) {
        offsets[nframes] = min(incl, max_rows);
        offsets[nframes + 1] = incl;
    }
}

// grid (blocks per frame, frames): the rows [offsets[f], offsets[f + 1]) of the packed arrays are the
// first rows of frame f's lists. 16-byte moves; a descriptor row is 32 of them.
__global__ __launch_bounds__(256) void k_pack_rows(const float4* __restrict__ kp, const float* __restrict__ orient,
                                                   const float4* __restrict__ desc, int cap,
                                                   const int* __restrict__ offsets, float4* __restrict__ out_kp,
                                                   float* __restrict__ out_orient, float4* __restrict__ out_desc)
{
    const int f = blockIdx.y;
    const int first = offsets[f];
    const int n = offsets[f + 1] - first;
    const size_t src = (size_t)f * cap;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (int i = tid; i < n; i += stride) {
        out_kp[first + i] = kp[src + i];
        if (out_orient) out_orient[first + i] = orient[src + i];
    }
    if (out_desc)
        for (int i = tid; i < n * 32; i += stride) out_desc[(size_t)first * 32 + i] = desc[src * 32 + i];
}

// One frame's descriptor rows and count into a caller's buffer (mif_sift_snapshot_frame).
__global__ __launch_bounds__(256) void k_snapshot_rows(const float4* __restrict__ desc, const int* __restrict__ count,
                                                       int cap, float4* __restrict__ out, int* __restrict__ out_count)
{
    const int n = min(*count, cap);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (int i = tid; i < n * 32; i += stride) out[i] = desc[i];
    if (tid == 0) *out_count = *count;
}

// ------------------------------------------------------------------ host side

struct OctaveBuf {
    int w = 0, h = 0, maxf = 0, words = 0, rows = 0;
    size_t items = 0;
    // every buffer holds `batch` frames back to back
    float* base = nullptr;                 // [h][w] input of level 0
    float* pyr = nullptr;                  // [6][h][w]
    float2* grad = nullptr;                // [3][h][w] gradient {mag, theta} of levels 1..3
    unsigned long long* mask = nullptr;    // [3*h][words] accepted positions
    unsigned long long* cmask = nullptr;   // [3*h][words] candidates (extrema above the contrast pre-test)
    int* rowcnt = nullptr;                 // [3*h]
    int* rowoff = nullptr;                 // [3*h]
};

// Build_separable_filter_cof, sift.cpp:819-889: sigma0 = 1.6, k = 2^(1/3), level
// s >= 1 uses the incremental sigma sqrt(sig_s^2 - sig_{s-1}^2), radius
// ceil(3.5 sigma), taps exp(-d^2 / (2 sigma^2)) normalised to sum 1.
void build_taps(std::vector<float> taps[kLevels], int fr[kLevels])
{
    const float sigma0 = 1.6f;
    const float k = powf(2.0f, 1.0f / kIntervals);
    float sig[kLevels];
    sig[0] = sigma0;
    for (int s = 1; s < kLevels; s++) {
        const float prev = sigma0 * powf(k, (float)(s - 1));
        const float next = prev * k;
        sig[s] = sqrtf(next * next - prev * prev);
    }
    for (int s = 0; s < kLevels; s++) {
        const size_t rad = (size_t)ceilf(3.5f * sig[s]);
        std::vector<float>& t = taps[s];
        t.assign(2 * rad + 1, 0.0f);
        float sum = 0.0f;
        for (size_t i = 0; i < rad; i++) {
            const size_t d = rad - i;
            const float q = (float)(d * d) / (sig[s] * sig[s]);
            const float v = expf((float)(-0.5 * q));
            t[i] = t[2 * rad - i] = v;
            sum += 2 * v;
        }
        t[rad] = 1.0f;
        sum += 1.0f;
        const float inv = 1.0f / sum;
        for (float& v : t) v *= inv;
        fr[s] = (int)rad;
    }
}

// Orientation bin of an angle exactly as the kernel of the reference computes it
// (sift.cpp:493, before the wrap of bin 36 to 0).
int ori_bin_of(float theta)
{
    return (int)mif_round((kOriBins * ((double)theta + K_M_PI)) / K_M_PI2);
}

// ori_bin_of is monotone in theta, so it is fully described by the smallest float that
// reaches each bin; they are found by bisection over the ordered bit patterns of
// [-float(pi), float(pi)], the range of the stored angles.
void build_ori_bins(OriBins* out)
{
    auto key_of = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    auto float_of = [](uint32_t k) { uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; memcpy(&f, &u, 4); return f; };
    out->t[0] = -INFINITY;
    out->t[kOriBins + 1] = INFINITY;
    for (int b = 1; b <= kOriBins; b++) {
        uint32_t lo = key_of(-MIF_PI_F), hi = key_of(MIF_PI_F);      // bin(lo) = 0 < b <= 36 = bin(hi)
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (ori_bin_of(float_of(mid)) >= b) hi = mid; else lo = mid;
        }
        out->t[b] = float_of(hi);
    }
}

} // namespace

struct mif_sift {
    int device = 0, W = 0, H = 0, C = 1, cap = 0, batch = 1;
    bool up = false;                       // first octave at twice the input size
    float* zero_row = nullptr;             // widest octave row of zeros: the blur's padding rows
    hipStream_t stream = nullptr;          // where the handle enqueues: its own stream unless mif_sift_set_stream
    hipStream_t own_stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_dep = nullptr;
    hipEvent_t ev_input = nullptr;         // the last batch's pixels have been read (first octave's scale space built)
    bool input_pending = false;
    OctaveBuf oct[kOctaves];
    uint8_t* d_pixels = nullptr;           // staging for the host-buffer entry point (one frame)
    float4* kp = nullptr;                  // [batch][cap]
    float* orient = nullptr;               // [batch][cap]
    float* desc = nullptr;                 // [batch][cap][128]
    int* counts = nullptr;                 // [batch][8]
    int* h_counts = nullptr;               // pinned, [batch][8]
    std::vector<float> taps[kLevels];
    int fr[kLevels];
    Taps ktaps[kLevels];
    OriBins ori_bins;
    bool frame_done = false;
    // optional per-stage timing: events at the stage boundaries of every octave
    bool timing = false, timed_frame = false, timed_desc = false;
    hipEvent_t ev_stage[kOctaves][3] = {};   // pyramid phase: before base, before blur, after blur
    hipEvent_t ev_feat[kOctaves][3] = {};    // feature phase: before detection, after emission, at the end
    int pyramid_frames = 0;                  // frames whose scale spaces await the feature phase
    int blur_launches = 0;                   // scale-space launches of the last enqueue_pyramid (all octaves)
    int desc_mode = 2;                       // k_descriptor<MODE>: 0 general loop, 1 streamlined walk, 2 mark and visit
    // streams and events for running the octaves' feature kernels side by side (small inputs)
    hipStream_t aux[kOctaves] = {};          // [0] unused: octave 0 stays on `stream`
    hipEvent_t ev_fork = nullptr, ev_scan[kOctaves] = {}, ev_join[kOctaves] = {}, ev_l3[kOctaves] = {};
    bool pyramid_split = false;              // the scale spaces in the handle were built on the side streams
    // the level-0 blur launches read the frames / the previous octave directly; base images are only
    // written on request (mif_sift_copy_base), from the pixels of the last call
    bool fused_base = false;
    const void* last_pixels = nullptr;
    // bit i: levels (2 i, 2 i + 1) of every octave are built by one launch (k_blur_pair). Only the first
    // pair by default: measured on 512 1080p frames, (0, 1) saves 3 % of the scale-space time, (2, 3) costs
    // 2 % and (4, 5) costs 8 % (larger windows: fewer resident blocks, more redundant halo arithmetic).
    unsigned pair_mask = 1u;
};

namespace {

// Whether the staging may fetch aligned groups of four pixels (stage_window4): the width a multiple of
// four, and every frame's first pixel on a 16-byte (float image) resp. 4-byte (8-bit frame) boundary.
// Used for BGR frames only: measured on 256 1080p frames it takes the BGR launch from 1.70 to 1.54 ms
// (one aligned 12-byte load in place of four unaligned 4-byte ones), while the float launches get slower
// (radius 6 / 7 / 9 / 11: 0.38 -> 0.40, 0.39 -> 0.42, 0.42 -> 0.49, 0.48 -> 0.56 ms) and so does the
// launch that reads 8-bit gray frames (32 4K frames: 0.80 -> 0.84 ms): their plain staging loads are
// coalesced already and cost a scalar row base each, the grouped form pays per-lane address and range
// arithmetic.
bool four_pixel_loads(const void* src, size_t src_stride, int w, int src_kind)
{
    static const bool off = getenv("MIF_NO_WIDE_STAGING") != nullptr;       // A/B switch
    if (off || w % 4 != 0 || w < 4) return false;
    const size_t unit = src_kind == kSrcFloat ? 4 : 1;                       // bytes per element of src_stride
    const size_t align = src_kind == kSrcFloat ? 16 : 4;
    return reinterpret_cast<uintptr_t>(src) % align == 0 && (src_stride * unit) % align == 0;
}

template <int FR, int SRC = kSrcFloat>
void launch_blur_fr(const void* src, size_t src_stride, float* dst, size_t pyr_stride, int w, int h, int nframes,
                    const Taps& t, const float* zero_row, hipStream_t st)
{
    const int tiles_x = (w + kBlurTile - 1) / kBlurTile, tiles_y = (h + kBlurTile - 1) / kBlurTile;
    const int ntiles = tiles_x * tiles_y * nframes;
    const int blocks = 8 * ((ntiles + 7) / 8);             // one equal share of the tiles per XCD
    if (SRC == kSrcBgrU8 && four_pixel_loads(src, src_stride, w, SRC))
        hipLaunchKernelGGL((k_blur_level<FR, SRC, true>), dim3(blocks), dim3(256), 0, st, src, src_stride, dst,
                           pyr_stride, w, h, t, tiles_x, tiles_y, ntiles, zero_row);
    else
        hipLaunchKernelGGL((k_blur_level<FR, SRC, false>), dim3(blocks), dim3(256), 0, st, src, src_stride, dst,
                           pyr_stride, w, h, t, tiles_x, tiles_y, ntiles, zero_row);
}

// Tap radius of level 0 (sigma 1.6): the only radius whose kernel exists for every source kind.
constexpr int kBaseRadius = 6;

int launch_blur(int fr, const void* src, size_t src_stride, float* dst, size_t pyr_stride, int w, int h, int nframes,
                const Taps& t, const float* zero_row, hipStream_t st, int src_kind = kSrcFloat)
{
    if (src_kind != kSrcFloat) {
        if (fr != kBaseRadius) { mif::set_error("no fused base-image blur for tap radius %d", fr); return MIF_ERR_UNSUPPORTED; }
        switch (src_kind) {
        case kSrcGrayU8: launch_blur_fr<kBaseRadius, kSrcGrayU8>(src, src_stride, dst, pyr_stride, w, h, nframes, t, zero_row, st); break;
        case kSrcBgrU8: launch_blur_fr<kBaseRadius, kSrcBgrU8>(src, src_stride, dst, pyr_stride, w, h, nframes, t, zero_row, st); break;
        default: launch_blur_fr<kBaseRadius, kSrcDecimated>(src, src_stride, dst, pyr_stride, w, h, nframes, t, zero_row, st); break;
        }
        return MIF_OK;
    }
    switch (fr) {
    case 5: launch_blur_fr<5>(src, src_stride, dst, pyr_stride, w, h, nframes, t, zero_row, st); break;
    case 6: launch_blur_fr<6>(src, src_stride, dst, pyr_stride, w, h, nframes, t, zero_row, st); break;
    case 7: launch_blur_fr<7>(src, src_stride, dst, pyr_stride, w, h, nframes, t, zero_row, st); break;
    case 9: launch_blur_fr<9>(src, src_stride, dst, pyr_stride, w, h, nframes, t, zero_row, st); break;
    case 11: launch_blur_fr<11>(src, src_stride, dst, pyr_stride, w, h, nframes, t, zero_row, st); break;
    default:
        mif::set_error("no blur kernel for tap radius %d", fr);
        return MIF_ERR_UNSUPPORTED;
    }
    return MIF_OK;
}

template <int FA, int FB, int SRC = kSrcFloat>
void launch_blur_pair_fr(const void* src, size_t src_stride, float* dst_a, float* dst_b, size_t pyr_stride, int w, int h,
                         int nframes, const Taps& ta, const Taps& tb, const float* zero_row, hipStream_t st)
{
    const int tiles_x = (w + kBlurTile - 1) / kBlurTile, tiles_y = (h + kBlurTile - 1) / kBlurTile;
    const int ntiles = tiles_x * tiles_y * nframes;
    const int blocks = 8 * ((ntiles + 7) / 8);             // one equal share of the tiles per XCD
    if (SRC == kSrcBgrU8 && four_pixel_loads(src, src_stride, w, SRC))
        hipLaunchKernelGGL((k_blur_pair<FA, FB, SRC, true>), dim3(blocks), dim3(256), 0, st, src, src_stride, dst_a,
                           dst_b, pyr_stride, w, h, ta, tb, tiles_x, tiles_y, ntiles, zero_row);
    else
        hipLaunchKernelGGL((k_blur_pair<FA, FB, SRC, false>), dim3(blocks), dim3(256), 0, st, src, src_stride, dst_a,
                           dst_b, pyr_stride, w, h, ta, tb, tiles_x, tiles_y, ntiles, zero_row);
}

// Two consecutive levels in one launch (k_blur_pair). Returns false when no kernel exists for the
// pair of tap radii / the source kind; the caller then launches the levels one by one.
bool launch_blur_pair(int fa, int fb, const void* src, size_t src_stride, float* dst_a, float* dst_b, size_t pyr_stride,
                      int w, int h, int nframes, const Taps& ta, const Taps& tb, const float* zero_row, hipStream_t st,
                      int src_kind)
{
#define MIF_PAIR(FA, FB, KIND)                                                                                        \
    launch_blur_pair_fr<FA, FB, KIND>(src, src_stride, dst_a, dst_b, pyr_stride, w, h, nframes, ta, tb, zero_row, st)
    if (fa == 6 && fb == 5) {
        switch (src_kind) {
        case kSrcFloat: MIF_PAIR(6, 5, kSrcFloat); return true;
        case kSrcGrayU8: MIF_PAIR(6, 5, kSrcGrayU8); return true;
        case kSrcBgrU8: MIF_PAIR(6, 5, kSrcBgrU8); return true;
        case kSrcDecimated: MIF_PAIR(6, 5, kSrcDecimated); return true;
        default: return false;
        }
    }
    if (src_kind != kSrcFloat) return false;
    if (fa == 6 && fb == 7) { MIF_PAIR(6, 7, kSrcFloat); return true; }
    if (fa == 9 && fb == 11) { MIF_PAIR(9, 11, kSrcFloat); return true; }
#undef MIF_PAIR
    return false;
}

// Writes the base image of octave o (the input of its level 0) for `nf` frames into O.base: the
// gray conversion of the frames (octave 0) or the decimated level 3 of the previous octave.
// `first` selects the frame of the batch to start from.
void write_base(mif_sift* s, int o, const void* d_pixels, int first, int nf, hipStream_t run)
{
    OctaveBuf& O = s->oct[o];
    const int items = (int)O.items;
    if (o == 0) {
        d_pixels = (const uint8_t*)d_pixels + (size_t)first * s->W * s->H * s->C;
        if (s->up) {
            // source size as the reference derives it (sift.cpp:1144-1145)
            const int ws = (int)((float)O.w / kResampleRatio), hs = (int)((float)O.h / kResampleRatio);
            const dim3 g((O.w + 255) / 256, O.h, nf);
            if (s->C == 3)
                hipLaunchKernelGGL(k_base_upscaled<3>, g, dim3(256), 0, run, (const uint8_t*)d_pixels, O.base, O.w,
                                   O.h, ws, hs, kResampleRatio);
            else
                hipLaunchKernelGGL(k_base_upscaled<1>, g, dim3(256), 0, run, (const uint8_t*)d_pixels, O.base, O.w,
                                   O.h, ws, hs, kResampleRatio);
        } else if (s->C == 3 && items % 4 == 0 && (reinterpret_cast<uintptr_t>(d_pixels) & 3u) == 0)
            hipLaunchKernelGGL(k_gray_bgr_x4, dim3((items / 4 + 255) / 256, 1, nf), dim3(256), 0, run,
                               (const uint8_t*)d_pixels, O.base, items);
        else if (s->C == 3)
            hipLaunchKernelGGL(k_gray_bgr, dim3((items + 255) / 256, 1, nf), dim3(256), 0, run,
                               (const uint8_t*)d_pixels, O.base, items);
        else if (items % 4 == 0 && (reinterpret_cast<uintptr_t>(d_pixels) & 3u) == 0)
            hipLaunchKernelGGL(k_gray_u8_x4, dim3((items / 4 + 255) / 256, 1, nf), dim3(256), 0, run,
                               (const uint8_t*)d_pixels, O.base, items);
        else
            hipLaunchKernelGGL(k_gray_u8, dim3((items + 255) / 256, 1, nf), dim3(256), 0, run,
                               (const uint8_t*)d_pixels, O.base, items);
    } else {
        const size_t prev_stride = s->oct[o - 1].items * kLevels;
        hipLaunchKernelGGL(k_decimate, dim3((O.w + 255) / 256, O.h, nf), dim3(256), 0, run,
                           (const float*)s->oct[o - 1].pyr + (size_t)first * prev_stride, prev_stride, O.base, O.w, O.h);
    }
}

// Phase 1 of a batch: base images and scale spaces (+ DoG) of all octaves. Only bandwidth
// kernels; the octaves depend on each other through the decimation.
// Little work (up to about one 1080p frame): every kernel is a chain of latencies, not a throughput
// problem, so the octaves run side by side on streams of their own where the data allow it.
bool small_input(const mif_sift* s, int nf)
{
    return !s->timing && s->aux[1] && (size_t)nf * s->oct[0].items <= ((size_t)1 << 21);
}

// `split`: the caller enqueues the feature phase right behind (enqueue_frames) and the input is small:
// octave o + 1 starts on its own stream as soon as level 3 of octave o exists (its base image is that
// level decimated) and the streams are joined at the end of the feature phase.
int enqueue_pyramid(mif_sift* s, const void* d_pixels, int nframes, bool split = false)
{
    hipStream_t st = s->stream;
    const int nf = nframes;
    MIF_HIP(hipEventRecord(s->ev0, st));
    MIF_HIP(hipMemsetAsync(s->counts, 0, (size_t)nf * CNT_WORDS * sizeof(int), st));
    const bool tm = s->timing;
    s->timed_frame = false;
    s->pyramid_frames = 0;
    s->pyramid_split = false;
    s->blur_launches = 0;
    if (split) {
        MIF_HIP(hipEventRecord(s->ev_fork, st));
        for (int o = 1; o < kOctaves; o++) MIF_HIP(hipStreamWaitEvent(s->aux[o], s->ev_fork, 0));
    }
    int last_l3 = -1;                           // latest octave whose level 3 has been enqueued
    bool input_recorded = false;                // ev_input marks the end of the first scale space enqueued
    for (int o = 0; o < kOctaves; o++) {
        OctaveBuf& O = s->oct[o];
        const int items = (int)O.items;
        if (items == 0) continue;
        hipStream_t run = (split && o > 0) ? s->aux[o] : s->stream;
        if (split && o > 0 && last_l3 >= 0) MIF_HIP(hipStreamWaitEvent(run, s->ev_l3[last_l3], 0));
        const size_t pyr_stride = O.items * kLevels;
        if (tm) MIF_HIP(hipEventRecord(s->ev_stage[o][0], st));
        // The level-0 launch forms the base image itself while it stages its window (see k_blur_level);
        // only the resampled first octave of up_scale is written out by a kernel of its own.
        const bool fused_base = s->fused_base && !(o == 0 && s->up);
        if (!fused_base) write_base(s, o, d_pixels, 0, nf, run);
        if (tm) MIF_HIP(hipEventRecord(s->ev_stage[o][1], st));
        // level 0 blurs the base image with sigma 1.6; level l >= 1 blurs level l-1
        // (sift.cpp:1196-1218); DoG[l-1] = level l - level l-1 (sift.cpp:256-260) is formed by its readers
        for (int l = 0; l < kLevels; l++) {
            const void* src = (l == 0) ? O.base : O.pyr + (size_t)(l - 1) * items;
            size_t src_stride = (l == 0) ? O.items : pyr_stride;
            int kind = kSrcFloat;
            if (l == 0 && fused_base) {
                if (o == 0) {
                    src = d_pixels;
                    src_stride = O.items * (size_t)s->C;
                    kind = s->C == 3 ? kSrcBgrU8 : kSrcGrayU8;
                } else {
                    src = s->oct[o - 1].pyr;
                    src_stride = s->oct[o - 1].items * kLevels;
                    kind = kSrcDecimated;
                }
            }
            float* dst = O.pyr + (size_t)l * items;
            // levels (0, 1), (2, 3), (4, 5) go two per launch where a kernel exists for the pair: the
            // first level of a pair is written but not read back (k_blur_pair)
            bool paired = false;
            if ((l & 1) == 0 && l + 1 < kLevels && (s->pair_mask >> (l / 2) & 1))
                paired = launch_blur_pair(s->fr[l], s->fr[l + 1], src, src_stride, dst, dst + items, pyr_stride, O.w, O.h,
                                          nf, s->ktaps[l], s->ktaps[l + 1], s->zero_row, run, kind);
            s->blur_launches++;
            if (paired) {
                l++;
            } else {
                const int rc = launch_blur(s->fr[l], src, src_stride, dst, pyr_stride, O.w, O.h, nf, s->ktaps[l],
                                           s->zero_row, run, kind);
                if (rc != MIF_OK) return rc;
            }
            if (split && l == kIntervals) {             // level 3 feeds the next octave
                MIF_HIP(hipEventRecord(s->ev_l3[o], run));
                last_l3 = o;
            }
        }
        if (tm) MIF_HIP(hipEventRecord(s->ev_stage[o][2], st));
        if (!input_recorded) {                      // nothing after this point reads the caller's pixels
            MIF_HIP(hipEventRecord(s->ev_input, run));
            s->input_pending = true;
            input_recorded = true;
        }
    }
    if (!input_recorded) {                          // no octave had a pixel: the call read nothing
        MIF_HIP(hipEventRecord(s->ev_input, st));
        s->input_pending = true;
    }
    MIF_HIP(hipGetLastError());
    s->pyramid_frames = nf;
    s->pyramid_split = split;
    s->last_pixels = d_pixels;
    return MIF_OK;
}

// Phase 2: detection, emission, gradients, orientations and descriptors of all octaves
// from the scale spaces phase 1 left in the handle.
int enqueue_features(mif_sift* s, int want_desc)
{
    hipStream_t st = s->stream;
    const int nf = s->pyramid_frames;
    const bool tm = s->timing;
    // Small inputs: the octaves run side by side on streams of their own. Only the row scans are
    // ordered octave after octave (they accumulate the list offsets the later kernels read).
    const bool side_by_side = small_input(s, nf);
    if (side_by_side && !s->pyramid_split) {
        MIF_HIP(hipEventRecord(s->ev_fork, st));
        for (int o = 1; o < kOctaves; o++) MIF_HIP(hipStreamWaitEvent(s->aux[o], s->ev_fork, 0));
    }
    if (!side_by_side && s->pyramid_split) {    // the scale spaces were built on the side streams
        for (int o = 1; o < kOctaves; o++) {
            MIF_HIP(hipEventRecord(s->ev_join[o], s->aux[o]));
            MIF_HIP(hipStreamWaitEvent(st, s->ev_join[o], 0));
        }
    }
    s->pyramid_split = false;
    int last_scanned = -1;                      // latest octave whose row scan has been enqueued
    for (int o = 0; o < kOctaves; o++) {
        OctaveBuf& O = s->oct[o];
        if (O.items == 0) continue;
        hipStream_t so = (side_by_side && o > 0) ? s->aux[o] : st;
        const size_t pyr_stride = O.items * kLevels;
        const size_t grad_stride = O.items * kIntervals;
        static const bool no_fuse = getenv("MIF_NO_FUSED_GRAD") != nullptr;      // A/B switch
        const bool fused_grad = want_desc && O.maxf > 0 && !no_fuse;
        if (tm) MIF_HIP(hipEventRecord(s->ev_feat[o][0], st));
        MIF_HIP(hipMemsetAsync(O.rowcnt, 0, (size_t)nf * O.rows * sizeof(int), so));
        MIF_HIP(hipMemsetAsync(O.cmask, 0, (size_t)nf * O.rows * O.words * sizeof(unsigned long long), so));
        {
            const int nbx = (O.w + 4 * kDetCols - 1) / (4 * kDetCols), nby = (O.h + kDetRows - 1) / kDetRows;
            const int nblocks = nbx * nby * nf;
            if (fused_grad)
                hipLaunchKernelGGL(k_detect<true>, dim3(8 * ((nblocks + 7) / 8)), dim3(256), 0, so, (const float*)O.pyr,
                                   pyr_stride, O.cmask, O.w, O.h, O.words, nbx, nby, nblocks, O.grad, grad_stride);
            else
                hipLaunchKernelGGL(k_detect<false>, dim3(8 * ((nblocks + 7) / 8)), dim3(256), 0, so, (const float*)O.pyr,
                                   pyr_stride, O.cmask, O.w, O.h, O.words, nbx, nby, nblocks, (float2*)nullptr, (size_t)0);
            const int nwords = nf * O.rows * O.words;
            hipLaunchKernelGGL(k_refine, dim3((nwords + 255) / 256), dim3(256), 0, so, (const float*)O.pyr, pyr_stride,
                               (const unsigned long long*)O.cmask, O.mask, O.rowcnt, O.w, O.h, O.words, o, nwords);
        }
        const bool grad_first = side_by_side && want_desc && O.maxf > 0 && !fused_grad;
        if (grad_first)                         // independent of the detector: ahead of the wait below
            hipLaunchKernelGGL(k_gradient, dim3((O.w + 127) / 128, (O.h + 2 * kGradRows - 1) / (2 * kGradRows), kIntervals * nf),
                               dim3(256), 0, so, (const float*)O.pyr, pyr_stride, O.grad, grad_stride, O.w, O.h);
        if (side_by_side && last_scanned >= 0) MIF_HIP(hipStreamWaitEvent(so, s->ev_scan[last_scanned], 0));
        hipLaunchKernelGGL(k_scan_rows, dim3(nf), dim3(1024), 0, so, (const int*)O.rowcnt, O.rowoff, O.rows, O.maxf,
                           o, s->counts);
        if (side_by_side) MIF_HIP(hipEventRecord(s->ev_scan[o], so));
        last_scanned = o;
        if (O.maxf > 0) {
            hipLaunchKernelGGL(k_emit, dim3((O.maxf + 255) / 256, nf), dim3(256), 0, so, (const float*)O.pyr, pyr_stride,
                               (const unsigned long long*)O.mask, (const int*)O.rowcnt, (const int*)O.rowoff, O.w,
                               O.h, O.words, o, O.maxf, s->cap, (const int*)s->counts, s->kp);
            if (tm) MIF_HIP(hipEventRecord(s->ev_feat[o][1], st));
            if (want_desc) {
                if (!grad_first && !fused_grad)
                    hipLaunchKernelGGL(k_gradient, dim3((O.w + 127) / 128, (O.h + 2 * kGradRows - 1) / (2 * kGradRows), kIntervals * nf),
                                       dim3(256), 0, so,
                                       (const float*)O.pyr, pyr_stride, O.grad, grad_stride, O.w, O.h);
                hipLaunchKernelGGL(k_orientation, dim3(std::min((O.maxf + 3) / 4, std::max(kOriBlocks, kOriBlocksTotal / nf)), nf), dim3(64), 0, so,
                                   (const float4*)s->kp,
                                   (const float2*)O.grad, grad_stride, O.w, O.h, o, s->cap, (const int*)s->counts,
                                   s->orient, s->ori_bins);
                const dim3 dgrid(std::min(O.maxf, std::max(kDescBlocks, kDescBlocksTotal / nf)), nf);
#define MIF_LAUNCH_DESC(MODE)                                                                                     \
    hipLaunchKernelGGL(k_descriptor<MODE>, dgrid, dim3(128), 0, so, (const float4*)s->kp, (const float*)s->orient, \
                       (const float2*)O.grad, grad_stride, O.w, O.h, o, s->cap, (const int*)s->counts, s->desc)
                if (s->desc_mode == 0) MIF_LAUNCH_DESC(0);
                else if (s->desc_mode == 1) MIF_LAUNCH_DESC(1);
                else MIF_LAUNCH_DESC(2);
#undef MIF_LAUNCH_DESC
            }
        } else if (tm) {
            MIF_HIP(hipEventRecord(s->ev_feat[o][1], st));
        }
        if (tm) MIF_HIP(hipEventRecord(s->ev_feat[o][2], st));
        if (side_by_side && o > 0) {
            MIF_HIP(hipEventRecord(s->ev_join[o], so));
            MIF_HIP(hipStreamWaitEvent(st, s->ev_join[o], 0));
        }
    }
    if (tm) { s->timed_frame = true; s->timed_desc = want_desc != 0; }
    hipLaunchKernelGGL(k_finish, dim3((nf + 63) / 64), dim3(64), 0, st, s->counts, nf, want_desc);
    MIF_HIP(hipGetLastError());
    MIF_HIP(hipEventRecord(s->ev1, st));
    s->frame_done = true;
    s->pyramid_frames = 0;
    return MIF_OK;
}

int enqueue_frames(mif_sift* s, const void* d_pixels, int nframes, int want_desc)
{
    // every cross-stream wait costs some microseconds: the early start of the next octave pays from
    // about half a million pixels on (measured: 0.81 -> 0.74 ms at 1080p, but +0.03 ms at 324 x 223)
    const bool split = small_input(s, nframes) && (size_t)nframes * s->oct[0].items >= ((size_t)1 << 19);
    const int rc = enqueue_pyramid(s, d_pixels, nframes, split);
    return rc != MIF_OK ? rc : enqueue_features(s, want_desc);
}

void free_all(mif_sift* s)
{
    for (int o = 0; o < kOctaves; o++) {
        OctaveBuf& O = s->oct[o];
        (void)hipFree(O.base); (void)hipFree(O.pyr); (void)hipFree(O.grad);
        (void)hipFree(O.mask); (void)hipFree(O.cmask); (void)hipFree(O.rowcnt); (void)hipFree(O.rowoff);
    }
    (void)hipFree(s->d_pixels); (void)hipFree(s->kp); (void)hipFree(s->orient);
    (void)hipFree(s->desc); (void)hipFree(s->counts); (void)hipFree(s->zero_row);
    if (s->h_counts) (void)hipHostFree(s->h_counts);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->ev_dep) (void)hipEventDestroy(s->ev_dep);
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    for (int o = 0; o < kOctaves; o++) {
        if (s->ev_scan[o]) (void)hipEventDestroy(s->ev_scan[o]);
        if (s->ev_join[o]) (void)hipEventDestroy(s->ev_join[o]);
        if (s->ev_l3[o]) (void)hipEventDestroy(s->ev_l3[o]);
        if (s->aux[o]) (void)hipStreamDestroy(s->aux[o]);
    }
    for (int o = 0; o < kOctaves; o++)
        for (int i = 0; i < 3; i++) {
            if (s->ev_stage[o][i]) (void)hipEventDestroy(s->ev_stage[o][i]);
            if (s->ev_feat[o][i]) (void)hipEventDestroy(s->ev_feat[o][i]);
        }
    if (s->ev_input) (void)hipEventDestroy(s->ev_input);
    if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
}

int create_impl(mif_sift* s)
{
    const size_t B = (size_t)s->batch;
    MIF_HIP(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
    s->stream = s->own_stream;
    MIF_HIP(hipEventCreate(&s->ev0));
    MIF_HIP(hipEventCreate(&s->ev1));
    MIF_HIP(hipEventCreateWithFlags(&s->ev_dep, hipEventDisableTiming));
    MIF_HIP(hipEventCreateWithFlags(&s->ev_input, hipEventDisableTiming));
    MIF_HIP(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
    for (int o = 0; o < kOctaves; o++) {
        // Side streams for the octaves of small inputs (see small_input): one-frame handles only. A
        // process has few hardware queues (4 by default) that all its streams share; a batched pipeline
        // already keeps two handles, a matcher and a copy stream busy, and every further stream would
        // end up behind another one's kernels.
        if (o > 0 && s->batch == 1) MIF_HIP(hipStreamCreateWithFlags(&s->aux[o], hipStreamNonBlocking));
        MIF_HIP(hipEventCreateWithFlags(&s->ev_scan[o], hipEventDisableTiming));
        MIF_HIP(hipEventCreateWithFlags(&s->ev_join[o], hipEventDisableTiming));
        MIF_HIP(hipEventCreateWithFlags(&s->ev_l3[o], hipEventDisableTiming));
    }
    s->cap = 0;
    for (int o = 0; o < kOctaves; o++) {
        OctaveBuf& O = s->oct[o];
        // sift.cpp:1313-1314: size_t = int / std::pow(2, id)
        if (s->up) {
            // sift.cpp:1306-1310: _resample_ratio (2.0f) * size_t / std::pow(2, id), truncated
            O.w = (int)(size_t)(kResampleRatio * (size_t)s->W / std::pow(2, o));
            O.h = (int)(size_t)(kResampleRatio * (size_t)s->H / std::pow(2, o));
        } else {
            O.w = (int)(size_t)(s->W / std::pow(2, o));
            O.h = (int)(size_t)(s->H / std::pow(2, o));
        }
        O.items = (size_t)O.w * O.h;
        O.maxf = (int)(O.items / 100);     // sift.cpp:1356
        O.words = (O.w + 63) / 64;
        O.rows = kIntervals * O.h;
        s->cap += O.maxf;
        if (O.items == 0) continue;
        MIF_HIP(hipMalloc(&O.pyr, B * O.items * kLevels * sizeof(float)));
        MIF_HIP(hipMalloc(&O.grad, B * O.items * kIntervals * sizeof(float2)));
        MIF_HIP(hipMalloc(&O.mask, B * O.rows * O.words * sizeof(unsigned long long)));
        MIF_HIP(hipMalloc(&O.cmask, B * O.rows * O.words * sizeof(unsigned long long)));
        MIF_HIP(hipMalloc(&O.rowcnt, B * O.rows * sizeof(int)));
        MIF_HIP(hipMalloc(&O.rowoff, B * O.rows * sizeof(int)));
    }
    if (s->cap < 1) s->cap = 1;            // keep every result buffer addressable
    const size_t cap = (size_t)s->cap;
    MIF_HIP(hipMalloc(&s->zero_row, (size_t)s->oct[0].w * sizeof(float)));
    MIF_HIP(hipMemset(s->zero_row, 0, (size_t)s->oct[0].w * sizeof(float)));
    MIF_HIP(hipMalloc(&s->d_pixels, (size_t)s->W * s->H * s->C));
    MIF_HIP(hipMalloc(&s->kp, B * cap * sizeof(float4)));
    MIF_HIP(hipMalloc(&s->orient, B * cap * sizeof(float)));
    MIF_HIP(hipMalloc(&s->desc, B * cap * 128 * sizeof(float)));
    MIF_HIP(hipMalloc(&s->counts, B * CNT_WORDS * sizeof(int)));
    MIF_HIP(hipMemset(s->counts, 0, B * CNT_WORDS * sizeof(int)));
    MIF_HIP(hipHostMalloc(&s->h_counts, B * CNT_WORDS * sizeof(int)));
    build_taps(s->taps, s->fr);
    build_ori_bins(&s->ori_bins);
    // a BGR pixel is fetched as a 32-bit word, which needs rows of at least four bytes
    if (const char* e = getenv("MIF_BLUR_PAIRS")) s->pair_mask = (unsigned)atoi(e) & 7u;      // A/B switch, 0 = one level per launch
    if (const char* e = getenv("MIF_DESC_PLAIN")) s->desc_mode = atoi(e) != 0 ? 0 : s->desc_mode;   // A/B switches
    if (const char* e = getenv("MIF_DESC_MODE")) s->desc_mode = std::max(0, std::min(2, atoi(e)));
    s->fused_base = s->fr[0] == kBaseRadius && s->oct[kOctaves - 1].w >= 2 && getenv("MIF_NO_FUSED_BASE") == nullptr;
    for (int o = 0; o < kOctaves; o++) {
        OctaveBuf& O = s->oct[o];
        if (O.items == 0) continue;
        // stored for the whole batch only where a kernel of its own writes it; one frame for the probe otherwise
        const bool stored = !s->fused_base || (s->up && o == 0);
        MIF_HIP(hipMalloc(&O.base, (stored ? B : 1) * O.items * sizeof(float)));
    }
    for (int l = 0; l < kLevels; l++) {
        memset(&s->ktaps[l], 0, sizeof(Taps));
        if ((int)s->taps[l].size() > kMaxTaps) {
            mif::set_error("filter of level %d has %zu taps, at most %d supported", l, s->taps[l].size(), kMaxTaps);
            return MIF_ERR_UNSUPPORTED;
        }
        for (size_t i = 0; i < s->taps[l].size(); i++) s->ktaps[l].t[i] = s->taps[l][i];
    }
    return MIF_OK;
}

int fetch_counts(mif_sift* s)
{
    MIF_HIP(hipMemcpyAsync(s->h_counts, s->counts, (size_t)s->batch * CNT_WORDS * sizeof(int), hipMemcpyDeviceToHost,
                           s->stream));
    MIF_HIP(hipStreamSynchronize(s->stream));
    return MIF_OK;
}

} // namespace

extern "C" {

int mif_sift_create_batched(int device, int width, int height, int channels, int up_scale, int batch,
                            mif_sift** out)
{
    if (!out) { mif::set_error("mif_sift_create: out is NULL"); return MIF_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    if (width <= 0 || height <= 0 || (channels != 1 && channels != 3) || batch < 1 || batch > 1024) {
        mif::set_error("mif_sift_create: bad geometry %dx%dx%d, batch %d", width, height, channels, batch);
        return MIF_ERR_INVALID_ARGUMENT;
    }
    const long long zoom = up_scale ? 2 : 1;
    if (zoom * width * zoom * height * 3 >= (1ll << 31) || zoom * height * batch * 3 > 65535ll * 64 ||
        zoom * height > 65535) {
        mif::set_error("mif_sift_create: image or batch too large");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(device);
    if (!guard.ok) { mif::set_error("mif_sift_create: cannot select device %d", device); return MIF_ERR_HIP; }
    mif_sift* s = new mif_sift();
    s->device = device; s->W = width; s->H = height; s->C = channels; s->batch = batch;
    s->up = up_scale != 0;
    const int rc = create_impl(s);
    if (rc != MIF_OK) { free_all(s); delete s; return rc; }
    *out = s;
    return MIF_OK;
}

int mif_sift_create(int device, int width, int height, int channels, int up_scale, mif_sift** out)
{
    return mif_sift_create_batched(device, width, height, channels, up_scale, 1, out);
}

int mif_sift_destroy(mif_sift* s)
{
    if (!s) return MIF_OK;
    mif::DeviceGuard guard(s->device);
    // a stream of the caller's (mif_sift_set_stream) may be gone already: wait for the device instead
    if (s->stream != s->own_stream) (void)hipDeviceSynchronize();
    else if (s->stream) (void)hipStreamSynchronize(s->stream);
    free_all(s);
    delete s;
    return MIF_OK;
}

int mif_sift_octave_info(const mif_sift* s, int octave, int* width, int* height, int* max_features)
{
    if (!s || octave < 0 || octave >= kOctaves) { mif::set_error("mif_sift_octave_info: bad argument"); return MIF_ERR_INVALID_ARGUMENT; }
    if (width) *width = s->oct[octave].w;
    if (height) *height = s->oct[octave].h;
    if (max_features) *max_features = s->oct[octave].maxf;
    return MIF_OK;
}

int mif_sift_capacity(const mif_sift* s) { return s ? s->cap : MIF_ERR_INVALID_ARGUMENT; }
int mif_sift_batch(const mif_sift* s) { return s ? s->batch : MIF_ERR_INVALID_ARGUMENT; }

int mif_sift_enqueue(mif_sift* s, const void* d_pixels, int nframes, int want_descriptors)
{
    if (!s || !d_pixels) { mif::set_error("mif_sift_enqueue: NULL argument"); return MIF_ERR_INVALID_ARGUMENT; }
    if (nframes < 1 || nframes > s->batch) {
        mif::set_error("mif_sift_enqueue: %d frames do not fit the handle's batch of %d", nframes, s->batch);
        return MIF_ERR_CAPACITY;
    }
    mif::DeviceGuard guard(s->device);
    return enqueue_frames(s, d_pixels, nframes, want_descriptors);
}

int mif_sift_enqueue_pyramid(mif_sift* s, const void* d_pixels, int nframes)
{
    if (!s || !d_pixels) { mif::set_error("mif_sift_enqueue_pyramid: NULL argument"); return MIF_ERR_INVALID_ARGUMENT; }
    if (nframes < 1 || nframes > s->batch) {
        mif::set_error("mif_sift_enqueue_pyramid: %d frames do not fit the handle's batch of %d", nframes, s->batch);
        return MIF_ERR_CAPACITY;
    }
    mif::DeviceGuard guard(s->device);
    return enqueue_pyramid(s, d_pixels, nframes);
}

int mif_sift_enqueue_features(mif_sift* s, int want_descriptors)
{
    if (!s) return MIF_ERR_INVALID_ARGUMENT;
    if (s->pyramid_frames < 1) {
        mif::set_error("mif_sift_enqueue_features: no scale space is pending (call mif_sift_enqueue_pyramid first)");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(s->device);
    return enqueue_features(s, want_descriptors);
}

int mif_sift_snapshot_frame(mif_sift* s, int frame, float* d_descriptors, int32_t* d_count)
{
    if (!s || !d_descriptors || !d_count || frame < 0 || frame >= s->batch) {
        mif::set_error("mif_sift_snapshot_frame: bad argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if (reinterpret_cast<uintptr_t>(d_descriptors) & 15u) {
        mif::set_error("mif_sift_snapshot_frame: the descriptor buffer must be 16-byte aligned");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(s->device);
    // only the rows the frame holds (a few thousand of a capacity of tens of thousands)
    hipLaunchKernelGGL(k_snapshot_rows, dim3(32), dim3(256), 0, s->stream,
                       (const float4*)(s->desc + (size_t)frame * s->cap * 128),
                       (const int*)(s->counts + (size_t)frame * CNT_WORDS + CNT_TOTAL), s->cap, (float4*)d_descriptors,
                       d_count);
    MIF_HIP(hipGetLastError());
    return MIF_OK;
}

int mif_sift_pack_results(mif_sift* s, int nframes, float* d_keypoints, float* d_orientations, float* d_descriptors,
                          int32_t* d_offsets, int max_rows)
{
    if (!s || !d_keypoints || !d_offsets || nframes < 1 || nframes > s->batch || max_rows < 0) {
        mif::set_error("mif_sift_pack_results: bad argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if ((d_orientations == nullptr) != (d_descriptors == nullptr)) {
        mif::set_error("mif_sift_pack_results: orientations and descriptors must both be given or both be NULL");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if ((reinterpret_cast<uintptr_t>(d_keypoints) | reinterpret_cast<uintptr_t>(d_descriptors)) & 15u) {
        mif::set_error("mif_sift_pack_results: keypoints and descriptors must be 16-byte aligned");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(s->device);
    hipLaunchKernelGGL(k_pack_offsets, dim3(1), dim3(1024), 0, s->stream, (const int*)s->counts, nframes, max_rows,
                       d_offsets);
    // a frame holds a few thousand rows of 32 16-byte pieces: 16 blocks of 256 threads walk them
    hipLaunchKernelGGL(k_pack_rows, dim3(16, nframes), dim3(256), 0, s->stream, (const float4*)s->kp,
                       (const float*)s->orient, (const float4*)s->desc, s->cap, (const int*)d_offsets,
                       (float4*)d_keypoints, d_orientations, (float4*)d_descriptors);
    MIF_HIP(hipGetLastError());
    return MIF_OK;
}

int mif_stream_wait_stream(int device, void* waiter, void* signaller)
{
    // Events are kept per calling thread and device and used in rotation (a call records one and makes
    // the waiter wait for that record; a later record of the same event does not change what an
    // earlier wait waits for), so the steady state creates nothing.
    constexpr int kDevices = 64, kRing = 16;
    struct Ring {
        hipEvent_t ev[kRing] = {};
        int next = 0;
    };
    static thread_local Ring rings[kDevices];
    if (device < 0 || device >= kDevices) {
        mif::set_error("mif_stream_wait_stream: device %d out of range", device);
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if (waiter == signaller) return MIF_OK;         // stream order already holds
    mif::DeviceGuard guard(device);
    if (!guard.ok) { mif::set_error("mif_stream_wait_stream: cannot select device %d", device); return MIF_ERR_HIP; }
    Ring& r = rings[device];
    hipEvent_t& ev = r.ev[r.next];
    r.next = (r.next + 1) % kRing;
    if (!ev) MIF_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, (hipStream_t)signaller);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
    if (e != hipSuccess) { mif::set_error("mif_stream_wait_stream: %s", hipGetErrorString(e)); return MIF_ERR_HIP; }
    return MIF_OK;
}

int mif_stream_create_masked(int device, const uint32_t* cu_mask, int words, void** stream)
{
    if (!cu_mask || words < 1 || !stream) {
        mif::set_error("mif_stream_create_masked: bad argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    // A mask of every other bit (0x55555555 in every word) made the runtime (ROCm 7.2) crash inside the
    // stream creation. Consecutive bits go to the 8 XCDs in turn, so that mask leaves XCDs 1, 3, 5, 7
    // without a compute unit; whether the empty XCDs or the gaps were the cause was not established
    // (the crashing call is not repeated to find out). The guard therefore admits only the family of
    // masks that has run: one contiguous run of set bits, within the device's compute units, that
    // leaves every XCD at least one (i.e. at least 8 bits long).
    constexpr int kXcds = 8;
    int first = -1, last = -1, set = 0;
    for (int i = 0; i < 32 * words; i++)
        if (cu_mask[i / 32] >> (i % 32) & 1u) {
            if (first < 0) first = i;
            last = i;
            set++;
        }
    if (set == 0 || last - first + 1 != set) {
        mif::set_error("mif_stream_create_masked: the mask must be one contiguous run of set bits");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if (set < kXcds) {
        mif::set_error("mif_stream_create_masked: %d compute units leave some of the %d XCDs without one", set, kXcds);
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(device);
    if (!guard.ok) { mif::set_error("mif_stream_create_masked: cannot select device %d", device); return MIF_ERR_HIP; }
    int ncu = 0;
    MIF_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
    if (last >= ncu) {
        mif::set_error("mif_stream_create_masked: bit %d is set, the device has %d compute units", last, ncu);
        return MIF_ERR_INVALID_ARGUMENT;
    }
    hipStream_t st = nullptr;
    MIF_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, cu_mask));
    *stream = st;
    return MIF_OK;
}

int mif_stream_destroy(int device, void* stream)
{
    if (!stream) return MIF_OK;
    mif::DeviceGuard guard(device);
    MIF_HIP(hipStreamSynchronize((hipStream_t)stream));
    MIF_HIP(hipStreamDestroy((hipStream_t)stream));
    return MIF_OK;
}

int mif_sift_set_stream(mif_sift* s, void* stream)
{
    if (!s) { mif::set_error("mif_sift_set_stream: NULL handle"); return MIF_ERR_INVALID_ARGUMENT; }
    mif::DeviceGuard guard(s->device);
    MIF_HIP(hipStreamSynchronize(s->stream));       // nothing of this handle is left on the old stream
    s->stream = stream ? (hipStream_t)stream : s->own_stream;
    return MIF_OK;
}

int mif_sift_wait_input_read(mif_sift* s, void* waiter)
{
    if (!s) { mif::set_error("mif_sift_wait_input_read: NULL handle"); return MIF_ERR_INVALID_ARGUMENT; }
    if (!s->input_pending) return MIF_OK;
    mif::DeviceGuard guard(s->device);
    MIF_HIP(hipStreamWaitEvent((hipStream_t)waiter, s->ev_input, 0));
    return MIF_OK;
}

int mif_sift_sync(mif_sift* s)
{
    if (!s) return MIF_ERR_INVALID_ARGUMENT;
    mif::DeviceGuard guard(s->device);
    MIF_HIP(hipStreamSynchronize(s->stream));
    return MIF_OK;
}

int mif_sift_wait_for(mif_sift* s, mif_sift* other)
{
    if (!s || !other) return MIF_ERR_INVALID_ARGUMENT;
    if (s == other) return MIF_OK;
    mif::DeviceGuard guard(s->device);
    MIF_HIP(hipEventRecord(other->ev_dep, other->stream));
    MIF_HIP(hipStreamWaitEvent(s->stream, other->ev_dep, 0));
    return MIF_OK;
}

int mif_sift_device_results(mif_sift* s, void** d_keypoints, void** d_orientations, void** d_descriptors,
                            void** d_counts)
{
    if (!s) return MIF_ERR_INVALID_ARGUMENT;
    if (d_keypoints) *d_keypoints = s->kp;
    if (d_orientations) *d_orientations = s->orient;
    if (d_descriptors) *d_descriptors = s->desc;
    if (d_counts) *d_counts = s->counts;
    return MIF_OK;
}

void* mif_sift_stream(mif_sift* s) { return s ? (void*)s->stream : nullptr; }

int mif_sift_read_counts(mif_sift* s, int frame, int counts[MIF_SIFT_OCTAVES], int raw_counts[MIF_SIFT_OCTAVES])
{
    if (!s || frame < 0 || frame >= s->batch) return MIF_ERR_INVALID_ARGUMENT;
    mif::DeviceGuard guard(s->device);
    const int rc = fetch_counts(s);
    if (rc != MIF_OK) return rc;
    const int* c = s->h_counts + (size_t)frame * CNT_WORDS;
    for (int o = 0; o < kOctaves; o++) {
        if (counts) counts[o] = c[CNT_OCT0 + o];
        if (raw_counts) raw_counts[o] = c[CNT_RAW0 + o];
    }
    return c[CNT_STATUS];
}

int mif_sift_extract(mif_sift* s, const uint8_t* pixels, float* keypoints, float* orientations,
                     float* descriptors, int capacity, int counts[MIF_SIFT_OCTAVES], float* elapsed_ms)
{
    if (!s || !pixels || !keypoints || !counts) {
        mif::set_error("mif_sift_extract: NULL argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if ((orientations == nullptr) != (descriptors == nullptr)) {
        mif::set_error("mif_sift_extract: orientations and descriptors must both be given or both be NULL");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(s->device);
    const int want_desc = descriptors ? 1 : 0;
    for (int o = 0; o < kOctaves; o++) counts[o] = 0;
    MIF_HIP(hipMemcpyAsync(s->d_pixels, pixels, (size_t)s->W * s->H * s->C, hipMemcpyHostToDevice, s->stream));
    int rc = enqueue_frames(s, s->d_pixels, 1, want_desc);
    if (rc != MIF_OK) return rc;
    rc = fetch_counts(s);
    if (rc != MIF_OK) return rc;
    if (elapsed_ms) MIF_HIP(hipEventElapsedTime(elapsed_ms, s->ev0, s->ev1));
    if (s->h_counts[CNT_STATUS] != 0) {
        mif::set_error("an octave produced no keypoints");
        return s->h_counts[CNT_STATUS];
    }
    const int total = s->h_counts[CNT_TOTAL];
    if (total > capacity) {
        mif::set_error("mif_sift_extract: %d keypoints do not fit capacity %d", total, capacity);
        return MIF_ERR_CAPACITY;
    }
    for (int o = 0; o < kOctaves; o++) counts[o] = s->h_counts[CNT_OCT0 + o];
    if (total > 0) {
        MIF_HIP(hipMemcpyAsync(keypoints, s->kp, (size_t)total * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
        if (want_desc) {
            MIF_HIP(hipMemcpyAsync(orientations, s->orient, (size_t)total * sizeof(float), hipMemcpyDeviceToHost,
                                   s->stream));
            MIF_HIP(hipMemcpyAsync(descriptors, s->desc, (size_t)total * 128 * sizeof(float),
                                   hipMemcpyDeviceToHost, s->stream));
        }
        MIF_HIP(hipStreamSynchronize(s->stream));
    }
    return MIF_OK;
}

int mif_sift_copy_results(mif_sift* s, int frame, float* keypoints, float* orientations, float* descriptors,
                          int capacity, int counts[MIF_SIFT_OCTAVES])
{
    if (!s || !keypoints || !counts || frame < 0 || frame >= s->batch) {
        mif::set_error("mif_sift_copy_results: bad argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(s->device);
    const int rc = fetch_counts(s);
    if (rc != MIF_OK) return rc;
    const int* c = s->h_counts + (size_t)frame * CNT_WORDS;
    for (int o = 0; o < kOctaves; o++) counts[o] = c[CNT_OCT0 + o];
    const int total = c[CNT_TOTAL];
    if (total > capacity) {
        mif::set_error("mif_sift_copy_results: %d keypoints do not fit capacity %d", total, capacity);
        return MIF_ERR_CAPACITY;
    }
    if (total > 0) {
        const size_t off = (size_t)frame * s->cap;
        MIF_HIP(hipMemcpy(keypoints, s->kp + off, (size_t)total * sizeof(float4), hipMemcpyDeviceToHost));
        if (orientations)
            MIF_HIP(hipMemcpy(orientations, s->orient + off, (size_t)total * sizeof(float), hipMemcpyDeviceToHost));
        if (descriptors)
            MIF_HIP(hipMemcpy(descriptors, s->desc + off * 128, (size_t)total * 128 * sizeof(float),
                              hipMemcpyDeviceToHost));
    }
    return c[CNT_STATUS];
}

int mif_sift_copy_level(mif_sift* s, int frame, int octave, int level, float* out)
{
    if (!s || !out || frame < 0 || frame >= s->batch || octave < 0 || octave >= kOctaves || level < 0 ||
        level >= kLevels) {
        mif::set_error("mif_sift_copy_level: bad argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(s->device);
    const OctaveBuf& O = s->oct[octave];
    MIF_HIP(hipStreamSynchronize(s->stream));
    if (O.items)
        MIF_HIP(hipMemcpy(out, O.pyr + ((size_t)frame * kLevels + level) * O.items, O.items * sizeof(float),
                          hipMemcpyDeviceToHost));
    return MIF_OK;
}

int mif_sift_copy_dog(mif_sift* s, int frame, int octave, int index, float* out)
{
    if (!s || !out || frame < 0 || frame >= s->batch || octave < 0 || octave >= kOctaves || index < 0 ||
        index >= kLevels - 1) {
        mif::set_error("mif_sift_copy_dog: bad argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(s->device);
    const OctaveBuf& O = s->oct[octave];
    MIF_HIP(hipStreamSynchronize(s->stream));
    if (O.items) {
        // the DoG images are not stored: two levels are fetched and subtracted here (one IEEE float
        // subtraction per pixel, kDiff of sift.cpp:256-260, as the detector does on the device)
        std::vector<float> lower(O.items);
        const float* base = O.pyr + ((size_t)frame * kLevels + index) * O.items;
        MIF_HIP(hipMemcpy(lower.data(), base, O.items * sizeof(float), hipMemcpyDeviceToHost));
        MIF_HIP(hipMemcpy(out, base + O.items, O.items * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < O.items; i++) out[i] = out[i] - lower[i];
    }
    return MIF_OK;
}

int mif_sift_copy_base(mif_sift* s, int frame, int octave, float* out)
{
    if (!s || !out || frame < 0 || frame >= s->batch || octave < 0 || octave >= kOctaves) {
        mif::set_error("mif_sift_copy_base: bad argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(s->device);
    const OctaveBuf& O = s->oct[octave];
    MIF_HIP(hipStreamSynchronize(s->stream));
    if (!O.items) return MIF_OK;
    const bool stored = !s->fused_base || (octave == 0 && s->up);
    if (!stored) {
        // not kept by the extraction: formed now from the pixels of the last call (octave 0) or from the
        // previous octave's scale space
        if (octave == 0 && !s->last_pixels) {
            mif::set_error("mif_sift_copy_base: no frame has been processed yet");
            return MIF_ERR_INVALID_ARGUMENT;
        }
        write_base(s, octave, s->last_pixels, frame, 1, s->stream);
        MIF_HIP(hipGetLastError());
        MIF_HIP(hipStreamSynchronize(s->stream));
    }
    MIF_HIP(hipMemcpy(out, O.base + (stored ? (size_t)frame * O.items : 0), O.items * sizeof(float),
                      hipMemcpyDeviceToHost));
    return MIF_OK;
}

// The reference's probes build their own device and tasks per call (sift.cpp:2103-2106, 2125-2129);
// here a call allocates its buffers, runs on the null stream and frees them.
static int probe_resize(int device, const uint8_t* pixels, int width, int height, int channels, float ratio_x,
                        float ratio_y, bool to_gray, void* out)
{
    const char* who = to_gray ? "mif_probe_gray_resize" : "mif_probe_resize_u8";
    if (!pixels || !out || width < 1 || height < 1 || !(ratio_x > 0.0f) || !(ratio_y > 0.0f) ||
        (to_gray ? channels != 3 : (channels != 1 && channels != 3))) {
        mif::set_error("%s: bad argument", who);
        return MIF_ERR_INVALID_ARGUMENT;
    }
    const int wd = (int)((float)width * ratio_x), hd = (int)((float)height * ratio_y);   // sift.cpp:2108-2109
    if (wd < 1 || hd < 1 || (double)wd * hd * 3 >= 2147483647.0 || (double)width * height * 3 >= 2147483647.0) {
        mif::set_error("%s: %d x %d -> %d x %d is out of range", who, width, height, wd, hd);
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(device);
    if (!guard.ok) { mif::set_error("%s: cannot select device %d", who, device); return MIF_ERR_HIP; }
    const size_t nsrc = (size_t)width * height, ndst = (size_t)wd * hd;
    uint8_t *d_src = nullptr, *d_dst8 = nullptr;
    float *d_gray = nullptr, *d_dstf = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_src); (void)hipFree(d_dst8); (void)hipFree(d_gray); (void)hipFree(d_dstf); };
    hipError_t e = hipMalloc(&d_src, nsrc * channels);
    if (e == hipSuccess) e = hipMemcpy(d_src, pixels, nsrc * channels, hipMemcpyHostToDevice);
    const dim3 grid((wd + 255) / 256, hd);
    if (to_gray) {
        if (e == hipSuccess) e = hipMalloc(&d_gray, nsrc * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&d_dstf, ndst * sizeof(float));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_gray_bgr, dim3((unsigned)((nsrc + 255) / 256)), dim3(256), 0, nullptr, d_src, d_gray,
                               (int)nsrc);
            hipLaunchKernelGGL((k_resize_nn<float, 1>), grid, dim3(256), 0, nullptr, d_dstf, (const float*)d_gray, wd,
                               hd, width, height, ratio_x, ratio_y);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpy(out, d_dstf, ndst * sizeof(float), hipMemcpyDeviceToHost);
    } else {
        if (e == hipSuccess) e = hipMalloc(&d_dst8, ndst * channels);
        if (e == hipSuccess) {
            if (channels == 3)
                hipLaunchKernelGGL((k_resize_nn<uint8_t, 3>), grid, dim3(256), 0, nullptr, d_dst8,
                                   (const uint8_t*)d_src, wd, hd, width, height, ratio_x, ratio_y);
            else
                hipLaunchKernelGGL((k_resize_nn<uint8_t, 1>), grid, dim3(256), 0, nullptr, d_dst8,
                                   (const uint8_t*)d_src, wd, hd, width, height, ratio_x, ratio_y);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpy(out, d_dst8, ndst * channels, hipMemcpyDeviceToHost);
    }
    cleanup();
    if (e != hipSuccess) { mif::set_error("%s: %s", who, hipGetErrorString(e)); return MIF_ERR_HIP; }
    return MIF_OK;
}

int mif_probe_resize_u8(int device, const uint8_t* pixels, int width, int height, int channels, float ratio_x,
                        float ratio_y, uint8_t* out)
{
    return probe_resize(device, pixels, width, height, channels, ratio_x, ratio_y, false, out);
}

int mif_probe_gray_resize(int device, const uint8_t* pixels, int width, int height, int channels, float ratio_x,
                          float ratio_y, float* out)
{
    return probe_resize(device, pixels, width, height, channels, ratio_x, ratio_y, true, out);
}

int mif_sift_taps(const mif_sift* s, int level, float* out, int capacity)
{
    if (!s || level < 0 || level >= kLevels) return MIF_ERR_INVALID_ARGUMENT;
    const int n = (int)s->taps[level].size();
    for (int i = 0; i < n && i < capacity; i++) out[i] = s->taps[level][i];
    return s->fr[level];
}

int mif_sift_set_timing(mif_sift* s, int enable)
{
    if (!s) return MIF_ERR_INVALID_ARGUMENT;
    mif::DeviceGuard guard(s->device);
    if (enable && !s->ev_stage[0][0]) {
        for (int o = 0; o < kOctaves; o++)
            for (int i = 0; i < 3; i++) {
                MIF_HIP(hipEventCreate(&s->ev_stage[o][i]));
                MIF_HIP(hipEventCreate(&s->ev_feat[o][i]));
            }
    }
    s->timing = enable != 0;
    return MIF_OK;
}

int mif_sift_stage_times(mif_sift* s, float ms[MIF_SIFT_OCTAVES][MIF_SIFT_STAGES])
{
    if (!s || !ms) return MIF_ERR_INVALID_ARGUMENT;
    if (!s->timed_frame) {
        mif::set_error("mif_sift_stage_times: no batch has been enqueued with timing enabled");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(s->device);
    MIF_HIP(hipStreamSynchronize(s->stream));
    for (int o = 0; o < kOctaves; o++)
        for (int i = 0; i < MIF_SIFT_STAGES; i++) {
            ms[o][i] = 0.0f;
            if (s->oct[o].items == 0) continue;
            // stages 0, 1: base image and scale space (pyramid phase); 2, 3: detector and
            // orientation + descriptor (feature phase)
            if (i < 2) MIF_HIP(hipEventElapsedTime(&ms[o][i], s->ev_stage[o][i], s->ev_stage[o][i + 1]));
            else MIF_HIP(hipEventElapsedTime(&ms[o][i], s->ev_feat[o][i - 2], s->ev_feat[o][i - 1]));
        }
    return MIF_OK;
}

int mif_sift_blur_launches(const mif_sift* s)
{
    return s ? s->blur_launches : MIF_ERR_INVALID_ARGUMENT;
}

int mif_sift_last_elapsed_ms(mif_sift* s, float* ms)
{
    if (!s || !ms || !s->frame_done) return MIF_ERR_INVALID_ARGUMENT;
    mif::DeviceGuard guard(s->device);
    MIF_HIP(hipEventSynchronize(s->ev1));
    MIF_HIP(hipEventElapsedTime(ms, s->ev0, s->ev1));
    return MIF_OK;
}

} // extern "C"
