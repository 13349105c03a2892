// GPU-resident SURF for MI355X (gfx950): integral image by wavefront prefix
// scans, Fast-Hessian response layers, 3-D non-maximum suppression with
// sub-pixel interpolation, Haar-wavelet orientation and 64-d descriptors.
//
// Replaces the reference's host pipeline (SURF/coopcl_surf/ccl_surf/ccl_surf.cpp)
// and its OpenCL kernels (SURF/coopcl_surf/CLSource/*.cl). Differences in
// structure, none in results:
//  * the integral image is a row scan followed by a column scan in place of
//    scan / transpose / scan / transpose (16 B per pixel instead of 32); both
//    scans keep the reference's association: Kogge-Stone over blocks of 64
//    elements plus the running total of the previous blocks;
//  * the 8 hessian_det launches are one launch, the 6 non_max_supression
//    launches are one launch that only marks accepted positions in a bitmask;
//    the interest points are then written in the fixed order (launch, row,
//    column) instead of atomic arrival order;
//  * the 109 orientation samples of a point sit in ascending work-item order.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "mif_common.h"
#include "mif_math.h"

namespace {

constexpr int kOctaves = 3;          // OCTAVES, surf_constants.h
constexpr int kSampleStep = 5;       // SAMPLE_STEP
constexpr float kThres = 0.0001f;    // THRES
constexpr int kMaxIpts = 8000;       // INIT_FEATS
constexpr int kLayers = 8;
constexpr int kNms = 6;              // non_max_supression launches: 3 octaves x 2
constexpr float kPiF = 3.14159274101257f;   // M_PI_F of OpenCL C

struct LayerDesc {
    int w, h, step, filter;
    size_t off;                      // offset of the layer in the response / laplacian pools
};

struct LayerTable {
    LayerDesc l[kLayers];
};

struct NmsDesc {
    int b, m, t;                     // layer indices (ccl_surf.cpp:109-113 filter_map)
    int row0;                        // first row of this launch in the row tables
};

struct NmsTable {
    NmsDesc n[kNms];
    int words;                       // bitmask words per row
    int rows;                        // total rows
};

// ccl_surf.cpp:743-762
__constant__ int c_tab_j[16] = {-12, -7, -2, 3, -12, -7, -2, 3, -12, -7, -2, 3, -12, -7, -2, 3};
__constant__ int c_tab_i[16] = {-12, -12, -12, -12, -7, -7, -7, -7, -2, -2, -2, -2, 3, 3, 3, 3};
__constant__ unsigned c_tab_id[13] = {6, 5, 4, 3, 2, 1, 0, 1, 2, 3, 4, 5, 6};
__constant__ float c_gauss25[49] = {
    0.02350693969273f, 0.01849121369071f, 0.01239503121241f, 0.00708015417522f, 0.00344628101733f, 0.00142945847484f, 0.00050524879060f,
    0.02169964028389f, 0.01706954162243f, 0.01144205592615f, 0.00653580605408f, 0.00318131834134f, 0.00131955648461f, 0.00046640341759f,
    0.01706954162243f, 0.01342737701584f, 0.00900063997939f, 0.00514124713667f, 0.00250251364222f, 0.00103799989504f, 0.00036688592278f,
    0.01144205592615f, 0.00900063997939f, 0.00603330940534f, 0.00344628101733f, 0.00167748505986f, 0.00069579213743f, 0.00024593098864f,
    0.00653580605408f, 0.00514124713667f, 0.00344628101733f, 0.00196854695367f, 0.00095819467066f, 0.00039744277546f, 0.00014047800980f,
    0.00318131834134f, 0.00250251364222f, 0.00167748505986f, 0.00095819467066f, 0.00046640341759f, 0.00019345616757f, 0.00006837798818f,
    0.00131955648461f, 0.00103799989504f, 0.00069579213743f, 0.00039744277546f, 0.00019345616757f, 0.00008024231247f, 0.00002836202103f};
// work items of getOrientationStep1 (13 x 13 grid) with i*i + j*j < 36, ascending
__constant__ unsigned char c_orient_item[109];
// Which of the 109 samples thread t of the orientation kernel evaluates: the samples sorted by (row,
// column) of the 13 x 13 grid, so that the lanes of a wave read along image rows (in work-item order the
// row changes from lane to lane and every lane's load lands in a cache line of its own). The results
// still go to the sample's own slot: the sums that follow run over the slots in work-item order.
__constant__ unsigned char c_orient_slot[109];

// ---------------------------------------------------------- integral image

// Gray value of one 8-bit pixel as the reference's host code produces it
// through OpenCV (ccl_surf.cpp:422-449): cvtColor BGR2GRAY fixed point
// (b*1868 + g*9617 + r*4899 + 8192) >> 14, then convertTo(CV_32F, 1/255).
// `last` is the index of the frame's last pixel: a BGR pixel is fetched as one (unaligned) 32-bit word
// that starts at its B byte, or one byte earlier for the frame's last pixel so that the word ends with
// the frame.
__device__ __forceinline__ uint32_t gray_fetch(const uint8_t* __restrict__ px, int i, int channels, int last)
{
    if (channels == 3) {
        uint32_t v;
        if (last >= 1) {
            const int off = (i < last) ? 3 * i : 3 * i - 1;
            __builtin_memcpy(&v, px + off, 4);
            v >>= 8 * (3 * i - off);
        } else {                                     // a frame of one pixel holds three bytes only
            v = px[0] | (px[1] << 8) | (px[2] << 16);
        }
        return v;
    }
    return px[i];
}

__device__ __forceinline__ float gray_value(uint32_t v, int channels)
{
    int g;
    if (channels == 3) {
        const int b = v & 255u, gg = (v >> 8) & 255u, r = (v >> 16) & 255u;
        g = (b * 1868 + gg * 9617 + r * 4899 + (1 << 13)) >> 14;
    } else {
        g = (int)v;
    }
    return (float)((double)g * (1.0 / 255.0));
}

__device__ __forceinline__ float gray_of(const uint8_t* __restrict__ px, int i, int channels, int last)
{
    return gray_value(gray_fetch(px, i, channels, last), channels);
}

// Inclusive scan over the 64 lanes of a wave with offsets 1, 2, ..., 32: the
// association tree of the reference's local-memory scan
// (integralImage_kernels.cl:120-141).
__device__ __forceinline__ float wave_scan64(float v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float u = __shfl_up(v, d);
        v = v + (lane >= d ? u : 0.0f);
    }
    return v;
}

// Row pass (integralImage_kernels.cl:63-152 `scan`, fused with the gray
// conversion): one wave per image row marches along it 64 pixels at a time.
// All kernels below take the frame of a batch from a grid dimension and address per-frame data at
// frame * stride.
// `only_if` (all three scan kernels; may be NULL): the launch does its work only when *only_if is not
// zero -- the launches that stand behind k_integral_fused and redo a batch whose bands gave up waiting.
__global__ __launch_bounds__(256) void k_scan_rows(const uint8_t* __restrict__ px, float* __restrict__ out, int w,
                                                   int h, int channels, const int* __restrict__ only_if)
{
    if (only_if && *only_if == 0) return;            // uniform over the grid
    constexpr int G = 8;                             // 64-pixel chunks whose loads are issued together
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= h) return;
    px += (size_t)blockIdx.y * w * h * channels;
    out += (size_t)blockIdx.y * w * h;
    float prev = 0.0f;
    for (int xg = 0; xg < w; xg += 64 * G) {
        float v[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int x = xg + 64 * g + lane;
            v[g] = (x < w) ? gray_of(px, y * w + x, channels, w * h - 1) : 0.0f;
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int x = xg + 64 * g + lane;
            if (xg + 64 * g < w) {                   // uniform over the wave
                const float sc = wave_scan64(v[g], lane);
                if (x < w) out[(size_t)y * w + x] = sc + prev;
                prev += __shfl(sc, 63);
            }
        }
    }
}

// (Two forms of the row pass that keep the tree in registers, a lane per row like k_scan_cols has a lane
// per column, were measured on 64 1080p BGR frames against the 315 us of the wave-per-row form above:
// each lane reading and writing its own row as 16-byte words took 305 us -- every instruction then
// touches 64 cache lines for 16 bytes each --, and a 64 x 64 tile transposed through LDS on the way in
// and out took 1143 us: 64 rows per wave leave 1088 waves for 64 frames, one per SIMD, each a long
// serial chain. The wave-per-row form has 69 120 waves to hide its cross-lane moves behind.)

// Column pass: the same scan along y (the reference transposes, scans rows and transposes back,
// ccl_surf.cpp:504-539): blocks of 64 rows are scanned with the Kogge-Stone tree and every block adds
// the running total of the blocks above it, prev_k = (((0 + t_0) + t_1) + ...) + t_(k-1), t = last
// element of a block's scan (integralImage_kernels.cl:143-150).
// A workgroup owns a strip of 64 columns of one frame, a wave one block of 64 rows of it (a lane = a
// column, so every access is a coalesced 256-byte row segment): the wave loads its 64 rows, runs the
// tree over them in registers (descending index, so every step reads values of the previous step) and
// publishes its total; after a barrier it sums the totals of the blocks above it in order, adds and
// stores. The trees of all row blocks of a strip thus run side by side; only the short chain of totals
// is sequential. Strips taller than 16 blocks take several rounds.
constexpr int kColWaves = 16;           // most waves of a workgroup
constexpr int kColMaxBlocks = 128;      // row blocks per strip the totals table holds (8192 rows)

// (Workgroups of 9 waves, so that the 17 row blocks of 1080 rows take two equally full rounds instead of
// 16 + 1, were measured: 415 instead of 368 us per 64 frames.)
__global__ __launch_bounds__(64 * kColWaves) void k_scan_cols(float* __restrict__ img, int w, int h,
                                                              const int* __restrict__ only_if)
{
    if (only_if && *only_if == 0) return;
    __shared__ float s_tot[kColMaxBlocks * 64];      // [row blocks][64]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = (int)(blockDim.x >> 6);
    const int x = blockIdx.x * 64 + lane;
    const bool live = x < w;
    img += (size_t)blockIdx.y * w * h;
    const int nblocks = (h + 63) / 64;
    for (int kb0 = 0; kb0 < nblocks; kb0 += nwaves) {        // uniform over the workgroup
        const int kb = kb0 + wave;
        const int y0 = kb * 64;
        float a[64];
        if (kb < nblocks) {
#pragma unroll
            for (int i = 0; i < 64; i++) a[i] = (live && y0 + i < h) ? img[(size_t)(y0 + i) * w + x] : 0.0f;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
                for (int i = 63; i >= d; i--) a[i] = a[i] + a[i - d];
            }
            s_tot[kb * 64 + lane] = a[63];
        }
        __syncthreads();
        if (kb < nblocks) {
            float prev = 0.0f;
            for (int j = 0; j < kb; j++) prev += s_tot[j * 64 + lane];
#pragma unroll
            for (int i = 0; i < 64; i++)
                if (live && y0 + i < h) img[(size_t)(y0 + i) * w + x] = a[i] + prev;
        }
    }
}

// The same pass with one lane per column walking all row blocks one after the other: for images taller
// than the tiled kernel's totals table.
__global__ __launch_bounds__(64) void k_scan_cols_serial(float* __restrict__ img, int w, int h,
                                                         const int* __restrict__ only_if)
{
    if (only_if && *only_if == 0) return;
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= w) return;
    img += (size_t)blockIdx.y * w * h;
    float prev = 0.0f;
    for (int y0 = 0; y0 < h; y0 += 64) {
        float a[64];
#pragma unroll
        for (int i = 0; i < 64; i++) a[i] = (y0 + i < h) ? img[(size_t)(y0 + i) * w + x] : 0.0f;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int i = 63; i >= d; i--) a[i] = a[i] + a[i - d];
        }
#pragma unroll
        for (int i = 0; i < 64; i++)
            if (y0 + i < h) img[(size_t)(y0 + i) * w + x] = a[i] + prev;
        prev += a[63];
    }
}

// Both passes in one launch: 3 (or 1) bytes read and 4 written per pixel instead of 3 + 4 + 4 + 4.
//
// A workgroup owns a band of 64 image rows of one frame, i.e. one row block of the column pass, and walks
// it from left to right a tile of 64 x 64 pixels, i.e. one column block of the row pass, at a time. Its
// four waves form a pipeline over the tiles, a barrier apart:
//   waves 0, 1  load tile s (a lane per column, coalesced), convert to gray, write it to LDS
//   wave 2      row pass of tile s - 1 with a lane per ROW: the tile's row in 64 registers, the tree in
//               registers (descending index, so every step reads values of the previous step), plus the
//               running total of the row's blocks to the left, which is one register of the lane; the
//               result goes back to LDS in place
//   wave 3      column pass of tile s - 2 with a lane per COLUMN, the same tree over the 64 rows; the total
//               of the row blocks above comes from the band above (below), is added, and the rows go to
//               memory as coalesced 256-byte segments
// so no step of either tree moves data between lanes. Three LDS tiles rotate through the three roles
// (pitch 65 floats: both the row-wise and the column-wise accesses are free of bank conflicts).
//
// The running total over the row blocks, prev_k = (((0 + t_0) + t_1) + ...) + t_(k-1), is handed down from
// band to band: wave 3 of band k publishes prev_(k+1) = prev_k + t_k for its 64 columns as soon as it has
// prev_k, as 64-bit words {value, launch number} written and polled with relaxed device-scope atomics
// (one word carries both, so no fence is needed; a word of an earlier launch has another number). The
// bands of a frame run one tile behind each other. Bands take their number from a ticket counter, frame
// major and top to bottom, so the band a workgroup waits for has always started before it, and the top
// band waits for nobody: every wait ends. The wait is bounded all the same; a band that gives up raises
// `fault`, goes on with zeros and waits no more, and the launch-per-pass kernels that stand behind this
// launch (they leave at once otherwise) then build the batch's integral images again.
constexpr int kFuPitch = 65;
constexpr int kFuSpin = 1 << 19;             // polls before a band gives up (a fraction of a second)
constexpr int kFusedMinBands = 192;     // bands of a batch from which the one-launch form is used (measured:
                                        // 1080p, 8 frames = 136 bands 3 % slower than two launches, 16 frames = 272 bands 3.5 % faster)

template <int CH, bool A4>
__global__ __launch_bounds__(256) void k_integral_fused(const uint8_t* __restrict__ px, float* __restrict__ out, int w,
                                                        int h, int nbands, int nitems,
                                                        unsigned long long* __restrict__ carry, int wpad,
                                                        unsigned int epoch, unsigned int epoch_wait,
                                                        unsigned int* __restrict__ ticket, int* __restrict__ fault)
{
    __shared__ float s_t[3][64 * kFuPitch];
    __shared__ int s_item;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    if (tid == 0) s_item = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int item = s_item;
    if (item >= nitems) return;                              // uniform over the workgroup
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = item / nbands, band = item - f * nbands;
    const int y0 = band * 64;
    px += (size_t)f * w * h * CH;
    out += (size_t)f * w * h;
    unsigned long long* carry_out = carry + ((size_t)f * nbands + band) * wpad;
    const unsigned long long* carry_in = carry_out - wpad;   // read for band > 0 only
    const int ncb = (w + 63) / 64;
    const int last = w * h - 1;                              // at least 1 (the host sends smaller frames elsewhere)
    const bool whole = y0 + 64 <= h;
    // Every role has a loop of its own (the registers of one role are then not live in the others), with
    // the same number of barriers: ncb + 2. A step's barrier waits for the wave's LDS operations only: a
    // __syncthreads() would also wait for the loads of the next tile and for the stores of the finished
    // one, a round trip to memory per step.
    auto step_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    if (wave < 2 && A4) {
        // Widths that are multiples of four: a lane owns four adjacent pixels of one row, 12 bytes of a BGR
        // frame (4 of a gray one) that start on a 4-byte boundary, and fetches them with one load; a wave's
        // load covers 4 rows x 64 columns. The raw words of the next tile are loaded a step ahead.
        constexpr int NW = CH == 3 ? 3 : 1;
        const int rr = lane >> 4, g = lane & 15;
        typedef uint32_t RawWords __attribute__((ext_vector_type(CH == 3 ? 3 : 1)));      // one load, one register tuple
        RawWords raw[8];
        auto fetch_tile = [&](int s) {
            // positions outside the image read the nearest group inside (and become zeros when they are
            // converted): no branch, all loads in flight together
            const int xc = min(s * 64 + 4 * g, w - 4);
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int yc = min(y0 + 32 * wave + 4 * it + rr, h - 1);
                raw[it] = *(const RawWords*)(px + ((size_t)yc * w + xc) * CH);
            }
        };
        fetch_tile(0);
        for (int s = 0; s < ncb + 2; s++) {
            if (s < ncb) {
                float* t = s_t[s % 3];
                const bool xin = s * 64 + 4 * g < w;
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const int row = 32 * wave + 4 * it + rr;
                    const bool in = xin && y0 + row < h;
                    uint32_t p[4];
                    if (CH == 3) {
                        const uint32_t w0 = raw[it][0], w1 = raw[it][NW > 1 ? 1 : 0], w2 = raw[it][NW > 2 ? 2 : 0];
                        p[0] = w0 & 0xffffffu;
                        p[1] = (w0 >> 24) | ((w1 & 0xffffu) << 8);
                        p[2] = (w1 >> 16) | ((w2 & 0xffu) << 16);
                        p[3] = w2 >> 8;
                    } else {
                        const uint32_t w0 = raw[it][0];
                        p[0] = w0 & 255u; p[1] = (w0 >> 8) & 255u; p[2] = (w0 >> 16) & 255u; p[3] = w0 >> 24;
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) t[row * kFuPitch + 4 * g + j] = in ? gray_value(p[j], CH) : 0.0f;
                }
                if (s + 1 < ncb) fetch_tile(s + 1);
            }
            step_barrier();
        }
    } else if (wave < 2) {
        // the raw pixels of the next tile are loaded a step ahead
        uint32_t raw[32];
        auto fetch_tile = [&](int s) {
            // positions outside the image read the nearest pixel inside (and become zeros when they are
            // converted): the loads then need no branch and are all in flight together
            const int xc = min(s * 64 + lane, w - 1);
            // (row addresses are scalars formed here, step by step: see the stores of wave 3)
            size_t pitch = (size_t)w * CH;
            asm volatile("" : "+s"(pitch));
            const int ya = y0 + 32 * wave;
            const uint8_t* row = px + (size_t)min(ya, h - 1) * w * CH;
#pragma unroll
            for (int r = 0; r < 32; r++) {
                const int yc = min(ya + r, h - 1);
                if (CH == 3) {  // one (unaligned) word from the pixel's B byte, one byte earlier for the last pixel
                    const int back = (yc * w + xc < last) ? 0 : 1;
                    uint32_t v;
                    __builtin_memcpy(&v, row + (3 * xc - back), 4);
                    raw[r] = v >> (8 * back);
                } else {
                    raw[r] = row[xc];
                }
                if (ya + r + 1 <= h - 1) row += pitch;
            }
        };
        fetch_tile(0);
        for (int s = 0; s < ncb + 2; s++) {
            if (s < ncb) {
                float* t = s_t[s % 3];
                const int x = s * 64 + lane;
#pragma unroll
                for (int r = 0; r < 32; r++) {
                    const int y = y0 + 32 * wave + r;
                    t[(32 * wave + r) * kFuPitch + lane] = (x < w && y < h) ? gray_value(raw[r], CH) : 0.0f;
                }
                if (s + 1 < ncb) fetch_tile(s + 1);
            }
            step_barrier();
        }
    } else if (wave == 2) {
        float prev_row = 0.0f;                               // total of this lane's row left of the tile
        for (int s = 0; s < ncb + 2; s++) {
            if (s >= 1 && s - 1 < ncb) {
                float* t = s_t[(s - 1) % 3] + lane * kFuPitch;
                float a[64];
#pragma unroll
                for (int j = 0; j < 64; j++) a[j] = t[j];
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
                    for (int j = 63; j >= d; j--) a[j] = a[j] + a[j - d];
                }
#pragma unroll
                for (int j = 0; j < 64; j++) t[j] = a[j] + prev_row;
                prev_row += a[63];
            }
            step_barrier();
        }
    } else {
        bool gave_up = false;
        for (int s = 0; s < ncb + 2; s++) {
            if (s >= 2) {
                const float* t = s_t[(s - 2) % 3];
                const int x = (s - 2) * 64 + lane;
                // the word of the band above is asked for before the tree and looked at after it
                unsigned long long v = 0ull;
                const bool chained = band > 0;
                if (chained) v = __hip_atomic_load(&carry_in[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float a[64];
#pragma unroll
                for (int i = 0; i < 64; i++) a[i] = t[i * kFuPitch + lane];
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
                    for (int i = 63; i >= d; i--) a[i] = a[i] + a[i - d];
                }
                float prev = 0.0f;
                if (chained) {
                    int tries = gave_up ? kFuSpin - 1 : 0;     // a band that gave up once does not wait again
                    while ((unsigned int)(v >> 32) != epoch_wait) {
                        if (++tries >= kFuSpin) { atomicExch(fault, 1); gave_up = true; v = 0ull; break; }
                        __builtin_amdgcn_s_sleep(1);
                        v = __hip_atomic_load(&carry_in[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    prev = __uint_as_float((unsigned int)v);
                }
                if (band + 1 < nbands)
                    __hip_atomic_store(&carry_out[x], ((unsigned long long)epoch << 32) | __float_as_uint(prev + a[63]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (x < w) {
                    // a scalar row address that moves down a row per store, plus the lane's column (the row
                    // pitch is hidden from the compiler, which would otherwise keep 64 row addresses in
                    // scalar registers across the loop and spill them)
                    size_t pitch = (size_t)w;
                    asm volatile("" : "+s"(pitch));
                    float* o = out + (size_t)y0 * w + (s - 2) * 64;
                    if (whole) {
#pragma unroll
                        for (int i = 0; i < 64; i++, o += pitch) o[lane] = a[i] + prev;
                    } else {
#pragma unroll
                        for (int i = 0; i < 64; i++, o += pitch)
                            if (y0 + i < h) o[lane] = a[i] + prev;
                    }
                }
            }
            step_barrier();
        }
    }
}

// Behind the launches that redo a batch after k_integral_fused gave up: counts the event and clears the
// mark, so that the stages downstream see a good batch.
__global__ void k_fault_handled(int* __restrict__ fault, int* __restrict__ fallbacks)
{
    if (*fault != 0) {
        *fallbacks += 1;
        *fault = 0;
    }
}

// ------------------------------------------------------------- box filters

// hessianDet_kernel.cl:77-103 BoxIntegral (same text in the orientation and
// descriptor kernel files)
__device__ __forceinline__ float box_integral(const float* __restrict__ data, int width, int height, int row, int col,
                                              int rows, int cols)
{
    float A = 0.0f, B = 0.0f, C = 0.0f, D = 0.0f;
    const int r1 = min(row, height) - 1;
    const int c1 = min(col, width) - 1;
    const int r2 = min(row + rows, height) - 1;
    const int c2 = min(col + cols, width) - 1;
    if (r1 >= 0 && c1 >= 0) A = data[r1 * width + c1];
    if (r1 >= 0 && c2 >= 0) B = data[r1 * width + c2];
    if (r2 >= 0 && c1 >= 0) C = data[r2 * width + c1];
    if (r2 >= 0 && c2 >= 0) D = data[r2 * width + c2];
    return fmaxf(0.0f, A - B - C + D);
}

__device__ __forceinline__ float haar_x(const float* __restrict__ img, int w, int h, int row, int column, int s)
{
    return box_integral(img, w, h, row - s / 2, column, s, s / 2) -
           box_integral(img, w, h, row - s / 2, column - s / 2, s, s / 2);
}

__device__ __forceinline__ float haar_y(const float* __restrict__ img, int w, int h, int row, int column, int s)
{
    return box_integral(img, w, h, row, column - s / 2, s / 2, s) -
           box_integral(img, w, h, row - s / 2, column - s / 2, s / 2, s);
}

// haar_x and haar_y of one sample together. For even s the four boxes have their corners on a 3 x 3 grid
// of integral-image positions (rows row - s/2, row, row + s/2; columns likewise), eight of which are
// used: they are loaded once instead of sixteen times, and every box is then the same expression
// max(0, A - B - C + D) on the same values as in box_integral.
__device__ __forceinline__ void haar_xy(const float* __restrict__ img, int w, int h, int row, int column, int s,
                                        float* hx, float* hy)
{
    if (s & 1) {                                     // not reached by the callers (s = 4 k or 2 k)
        *hx = haar_x(img, w, h, row, column, s);
        *hy = haar_y(img, w, h, row, column, s);
        return;
    }
    const int half = s / 2;
    const int r0 = min(row - half, h) - 1, r1 = min(row, h) - 1, r2 = min(row + half, h) - 1;
    const int c0 = min(column - half, w) - 1, c1 = min(column, w) - 1, c2 = min(column + half, w) - 1;
    auto at = [&](int r, int c) { return (r >= 0 && c >= 0) ? img[r * w + c] : 0.0f; };
    const float p00 = at(r0, c0), p01 = at(r0, c1), p02 = at(r0, c2);
    const float p10 = at(r1, c0), p12 = at(r1, c2);
    const float p20 = at(r2, c0), p21 = at(r2, c1), p22 = at(r2, c2);
    *hx = fmaxf(0.0f, p01 - p02 - p21 + p22) - fmaxf(0.0f, p00 - p01 - p20 + p21);
    *hy = fmaxf(0.0f, p10 - p12 - p20 + p22) - fmaxf(0.0f, p00 - p02 - p10 + p12);
}

// haar_xy for small boxes (half = s / 2 of 2 or 3, the sizes of the finest two scales): the three corners
// of a grid row lie within five resp. seven values, so the first two come with ONE 16-byte load (any
// 4-byte boundary) and a row costs two cache accesses instead of three. Positions whose corner grid
// touches the image border take the general form. Same values, same expressions.
__device__ __forceinline__ void haar_xy_small(const float* __restrict__ img, int w, int h, int row, int column, int s,
                                              float* hx, float* hy)
{
    const int half = s / 2;
    const int ra = row - half - 1, ca = column - half - 1;          // first corner
    if (ra < 0 || ca < 0 || row + half > h || column + half > w || ca + 3 >= w) {
        haar_xy(img, w, h, row, column, s, hx, hy);
        return;
    }
    typedef float F4 __attribute__((ext_vector_type(4), aligned(4)));
    const float* q0 = img + ra * w + ca;
    const float* q1 = q0 + half * w;
    const float* q2 = q1 + half * w;
    const F4 a = *(const F4*)q0, b = *(const F4*)q1, c = *(const F4*)q2;
    const float p00 = a[0], p01 = half == 2 ? a[2] : a[3], p02 = q0[2 * half];
    const float p10 = b[0], p12 = q1[2 * half];
    const float p20 = c[0], p21 = half == 2 ? c[2] : c[3], p22 = q2[2 * half];
    *hx = fmaxf(0.0f, p01 - p02 - p21 + p22) - fmaxf(0.0f, p00 - p01 - p20 + p21);
    *hy = fmaxf(0.0f, p10 - p12 - p20 + p22) - fmaxf(0.0f, p00 - p02 - p10 + p12);
}

// hessianDet_kernel.cl:105-175 for all eight layers in one launch
// (grid.z = layer). Work items beyond the layer do not write (the reference's
// guard only checks the image bounds and writes one element past the row).
__global__ __launch_bounds__(256) void k_hessian(const float* __restrict__ img, int width, int height, LayerTable tab,
                                                 float* __restrict__ responses, int* __restrict__ laplacians,
                                                 size_t pool, int first_layer)
{
    // grid.z = (layers first_layer .. kLayers - 1) x frames
    const int nlayers = kLayers - first_layer;
    const int frame = blockIdx.z / nlayers;
    const LayerDesc L = tab.l[first_layer + blockIdx.z - frame * nlayers];
    img += (size_t)frame * width * height;
    responses += (size_t)frame * pool;
    laplacians += (size_t)frame * pool;
    const int idx = blockIdx.x * 16 + (threadIdx.x & 15);
    const int idy = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (idx >= L.w || idy >= L.h) return;
    const int w = L.filter;
    const int l = L.filter / 3;
    const int b = (L.filter - 1) / 2 + 1;
    const float inverse_area = 1.0f / (w * w);
    const int r = idy * L.step;
    const int c = idx * L.step;
    if (r >= height || c >= width) return;
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
    responses[L.off + idy * L.w + idx] = (Dxx * Dyy - 0.81f * Dxy * Dxy);
    laplacians[L.off + idy * L.w + idx] = (Dxx + Dyy >= 0 ? 1 : 0);
}

// The four layers of the first octave (filters 9, 15, 21, 27) share the sampling step and the output
// grid, and together hold 87 % of all responses at 1080p. Read straight from memory every response costs
// 32 scattered loads (a vector-cache access per lane and load: the launch is bound by those). Here a block
// stages the part of the integral image that its 16 x 16 output positions touch for the largest of the
// filters -- (16 - 1) step + 1 pixels extended by the filter's reach, 104 x 104 values at step 5 -- into
// LDS once, with coalesced row loads and zeros where the row or column index is negative (what
// BoxIntegral returns for those corners), and evaluates the four layers from there: the same corner
// values, the same expressions, hence the same bits. Lanes run along x with a stride of `step` values,
// which is odd, so the LDS reads of a row of outputs are conflict-free.
constexpr int kHessTileOut = 16;
constexpr int kHessFirstLayers = 4;              // layers served by the staged form
constexpr int kHessMaxFilter = 27;
constexpr int kHessReachLo = (kHessMaxFilter - 1) / 2 + 1 + 1;   // b + 1 values before the sample position
constexpr int kHessReachHi = (kHessMaxFilter - 1) / 2;           // b - 1 values behind it
constexpr int kHessWin = (kHessTileOut - 1) * kSampleStep + 1 + kHessReachLo + kHessReachHi;   // 104
constexpr int kHessPitch = kHessWin + 1;

// BoxIntegral (hessianDet_kernel.cl:77-103) on the staged window: (wy0, wx0) is the image position of
// the window's first value; clamped indices below zero meet the zeros stored there.
__device__ __forceinline__ float box_integral_staged(const float* __restrict__ s_win, int wy0, int wx0, int width,
                                                     int height, int row, int col, int rows, int cols)
{
    const int r1 = min(row, height) - 1 - wy0;
    const int c1 = min(col, width) - 1 - wx0;
    const int r2 = min(row + rows, height) - 1 - wy0;
    const int c2 = min(col + cols, width) - 1 - wx0;
    const float A = s_win[r1 * kHessPitch + c1];
    const float B = s_win[r1 * kHessPitch + c2];
    const float C = s_win[r2 * kHessPitch + c1];
    const float D = s_win[r2 * kHessPitch + c2];
    return fmaxf(0.0f, A - B - C + D);
}

// grid (ceil(layer w / 16), ceil(layer h / 16), frames); the layers 0 .. kHessFirstLayers - 1 must share
// w, h and step = kSampleStep and have filters of at most kHessMaxFilter (checked by the host).
__global__ __launch_bounds__(256) void k_hessian_staged(const float* __restrict__ img, int width, int height,
                                                        LayerTable tab, float* __restrict__ responses,
                                                        int* __restrict__ laplacians, size_t pool)
{
    __shared__ float s_win[kHessWin * kHessPitch];
    const int frame = blockIdx.z;
    img += (size_t)frame * width * height;
    responses += (size_t)frame * pool;
    laplacians += (size_t)frame * pool;
    const int tid = threadIdx.x;
    const int wx0 = (int)blockIdx.x * kHessTileOut * kSampleStep - kHessReachLo;
    const int wy0 = (int)blockIdx.y * kHessTileOut * kSampleStep - kHessReachLo;
    {
        // A thread owns one column of every second row of the window (columns 104 .. 127 of the thread
        // grid idle): the column's clamped address and its inside test are formed once, a turn costs the
        // row's clamp, one load and one store. The loads of sixteen turns are issued before their LDS stores
        // (clamped addresses, zeros selected afterwards), so that their latencies overlap.
        const int lx = tid & 127, ty = tid >> 7;
        const int gx = wx0 + lx;
        const bool inx = gx >= 0 && gx < width;
        const float* col = img + min(max(gx, 0), width - 1);
        constexpr int kTurns = (kHessWin + 1) / 2, kGroup = 13;
        static_assert(kTurns % kGroup == 0, "turns come in whole groups");
        if (lx < kHessWin) {
            for (int g0 = 0; g0 < kTurns; g0 += kGroup) {
                float v[kGroup];
#pragma unroll
                for (int it = 0; it < kGroup; it++) {
                    const int gy = wy0 + ty + 2 * (g0 + it);
                    v[it] = col[min(max(gy, 0), height - 1) * width];
                }
#pragma unroll
                for (int it = 0; it < kGroup; it++) {
                    const int ly = ty + 2 * (g0 + it);
                    const int gy = wy0 + ly;
                    s_win[ly * kHessPitch + lx] = (inx && gy >= 0 && gy < height) ? v[it] : 0.0f;
                }
            }
        }
    }
    __syncthreads();
    const int idx = blockIdx.x * kHessTileOut + (tid & 15);
    const int idy = blockIdx.y * kHessTileOut + (tid >> 4);
    const int lw = tab.l[0].w, lh = tab.l[0].h;
    if (idx >= lw || idy >= lh) return;
    const int r = idy * kSampleStep;
    const int c = idx * kSampleStep;
    if (r >= height || c >= width) return;
#pragma unroll
    for (int li = 0; li < kHessFirstLayers; li++) {
        const LayerDesc L = tab.l[li];
        const int w = L.filter;
        const int l = L.filter / 3;
        const int b = (L.filter - 1) / 2 + 1;
        const float inverse_area = 1.0f / (w * w);
#define MIF_BOX(row, col, rows, cols) box_integral_staged(s_win, wy0, wx0, width, height, row, col, rows, cols)
        float Dxx = MIF_BOX(r - l + 1, c - b, 2 * l - 1, w) - MIF_BOX(r - l + 1, c - l / 2, 2 * l - 1, l) * 3;
        float Dyy = MIF_BOX(r - b, c - l + 1, w, 2 * l - 1) - MIF_BOX(r - l / 2, c - l + 1, l, 2 * l - 1) * 3;
        float Dxy = MIF_BOX(r - l, c + 1, l, l) + MIF_BOX(r + 1, c - l, l, l) - MIF_BOX(r - l, c - l, l, l) -
                    MIF_BOX(r + 1, c + 1, l, l);
#undef MIF_BOX
        Dxx *= inverse_area;
        Dyy *= inverse_area;
        Dxy *= inverse_area;
        responses[L.off + idy * L.w + idx] = (Dxx * Dyy - 0.81f * Dxy * Dxy);
        laplacians[L.off + idy * L.w + idx] = (Dxx + Dyy >= 0 ? 1 : 0);
    }
}

// ------------------------------------------------------------------- NMS

struct LayerView {
    const float* resp;
    const int* lap;
    int w, h, step, filter;
};

__device__ __forceinline__ LayerView view_of(const LayerTable& tab, int i, const float* responses, const int* laps)
{
    LayerView v;
    v.resp = responses + tab.l[i].off;
    v.lap = laps + tab.l[i].off;
    v.w = tab.l[i].w; v.h = tab.l[i].h; v.step = tab.l[i].step; v.filter = tab.l[i].filter;
    return v;
}

__device__ __forceinline__ float resp_at(const LayerView& L, int c, int r, int scale)
{
    return L.resp[r * scale * L.w + c * scale];
}

// nonMaxSuppression_kernel.cl:232-296 isExtremum
__device__ bool is_extremum(const LayerView& t, const LayerView& m, const LayerView& b, int c, int r, float threshold)
{
    const int layerBorder = (t.filter + 1) / (2 * t.step);
    if (r <= layerBorder || r >= t.h - layerBorder || c <= layerBorder || c >= t.w - layerBorder) return false;
    const int mScale = m.w / t.w;
    const float candidate = resp_at(m, c, r, mScale);
    if (candidate < threshold) return false;
    const int bScale = b.w / t.w;
    float localMax = resp_at(t, c - 1, r - 1, 1);
#pragma unroll
    for (int dr = -1; dr <= 1; dr++)
#pragma unroll
        for (int dc = -1; dc <= 1; dc++) {
            localMax = fmaxf(localMax, resp_at(t, c + dc, r + dr, 1));
            localMax = fmaxf(localMax, resp_at(b, c + dc, r + dr, bScale));
            if (dr != 0 || dc != 0) localMax = fmaxf(localMax, resp_at(m, c + dc, r + dr, mScale));
        }
    return !(localMax > candidate);
}

struct IptCore {
    float x, y, scale;
    int laplacian;
};

// nonMaxSuppression_kernel.cl:107-228 interpolateExtremum
__device__ bool interpolate_extremum(const LayerView& t, const LayerView& m, const LayerView& b, int r, int c,
                                     IptCore* out)
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
    if (mif_fabsf(xi) < 0.5f && mif_fabsf(xr) < 0.5f && mif_fabsf(xc) < 0.5f) {
        const int filterStep = m.filter - b.filter;
        out->x = ((float)c + xc) * (float)t.step;
        out->y = ((float)r + xr) * (float)t.step;
        out->scale = 0.1333f * ((float)m.filter + (xi * (float)filterStep));
        out->laplacian = m.lap[(r * mScale) * m.w + c * mScale];
        return true;
    }
    return false;
}

// nonMaxSuppression_kernel.cl:306-364 for the six (bottom, middle, top)
// triples in one launch: grid (words, ceil(rows / 4), frames) over the rows of all six middle layers laid
// end to end (NmsDesc::row0), a wave per row and word of 64 columns. The work items of a launch of the
// reference span the middle layer; (c, r) are read as top-layer coordinates. A wave stores its ballot of
// accepted positions as one bitmask word and adds its count to the row total.
__global__ __launch_bounds__(256) void k_nms_mark(LayerTable tab, NmsTable nms, const float* __restrict__ responses,
                                                  const int* __restrict__ laps, float threshold,
                                                  unsigned long long* __restrict__ mask, int* __restrict__ rowcnt,
                                                  size_t pool)
{
    const int frame = blockIdx.z;
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= nms.rows) return;                         // uniform over the wave
    int li = 0;
    for (int i = 1; i < kNms; i++)
        if (row >= nms.n[i].row0) li = i;
    const NmsDesc d = nms.n[li];
    responses += (size_t)frame * pool;
    laps += (size_t)frame * pool;
    mask += (size_t)frame * nms.rows * nms.words;
    rowcnt += (size_t)frame * nms.rows;
    const LayerView b = view_of(tab, d.b, responses, laps);
    const LayerView m = view_of(tab, d.m, responses, laps);
    const LayerView t = view_of(tab, d.t, responses, laps);
    const int r = row - d.row0;
    const int c = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= m.w) return;                  // uniform over the wave
    bool found = false;
    if (c < m.w && t.w > 0 && t.h > 0 && is_extremum(t, m, b, c, r, threshold)) {
        IptCore p;
        found = interpolate_extremum(t, m, b, r, c, &p);
    }
    const unsigned long long bits = __ballot(found);
    if (lane == 0) {
        mask[(size_t)row * nms.words + blockIdx.x] = bits;
        if (bits) atomicAdd(&rowcnt[row], __popcll(bits));
    }
}

// Exclusive scan of the row totals, point count cut at maxIpts. One block per frame.
__global__ __launch_bounds__(1024) void k_scan_counts(const int* __restrict__ rowcnt, int* __restrict__ rowoff,
                                                      int rows, int max_ipts, int* __restrict__ counts,
                                                      const int* __restrict__ fault)
{
    __shared__ int s_part[1024];
    const int tid = threadIdx.x;
    rowcnt += (size_t)blockIdx.x * rows;
    rowoff += (size_t)blockIdx.x * rows;
    counts += (size_t)blockIdx.x * 2;
    const int per = (rows + 1023) / 1024;
    const int lo = min(rows, tid * per), hi = min(rows, lo + per);
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
        // a batch whose integral images are not to be trusted (k_integral_fused) has no points: raw total -2
        const bool bad = *fault != 0;
        counts[0] = bad ? 0 : min(s_part[1023], max_ipts);
        counts[1] = bad ? -2 : s_part[1023];
    }
}

// Writes the accepted positions in (launch, row, column) order: one wave per
// row; a set bit's rank in the row gives its slot.
__global__ __launch_bounds__(256) void k_nms_emit(LayerTable tab, NmsTable nms, const float* __restrict__ responses,
                                                  const int* __restrict__ laps,
                                                  const unsigned long long* __restrict__ mask,
                                                  const int* __restrict__ rowcnt, const int* __restrict__ rowoff,
                                                  int max_ipts, float2* __restrict__ pos, float* __restrict__ scale,
                                                  int* __restrict__ lap_out, size_t pool)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nms.rows) return;
    {
        const size_t f = blockIdx.y;
        responses += f * pool;
        laps += f * pool;
        mask += f * nms.rows * nms.words;
        rowcnt += f * nms.rows;
        rowoff += f * nms.rows;
        pos += f * max_ipts;
        scale += f * max_ipts;
        lap_out += f * max_ipts;
    }
    if (rowcnt[row] == 0) return;
    int li = 0;
    for (int i = 1; i < kNms; i++)
        if (row >= nms.n[i].row0) li = i;
    const NmsDesc d = nms.n[li];
    const LayerView b = view_of(tab, d.b, responses, laps);
    const LayerView m = view_of(tab, d.m, responses, laps);
    const LayerView t = view_of(tab, d.t, responses, laps);
    const int r = row - d.row0;
    int slot = rowoff[row];
    for (int wd = 0; wd < nms.words; wd++) {
        const unsigned long long bits = mask[(size_t)row * nms.words + wd];
        if (bits == 0) continue;
        if ((bits >> lane) & 1ull) {
            const int p = slot + __popcll(bits & ((1ull << lane) - 1ull));
            if (p < max_ipts) {
                IptCore ip;
                interpolate_extremum(t, m, b, r, wd * 64 + lane, &ip);
                pos[p] = make_float2(ip.x, ip.y);
                scale[p] = ip.scale;
                lap_out[p] = ip.laplacian;
            }
        }
        slot += __popcll(bits);
    }
}

// ------------------------------------------------------------ orientation

// How the orientation and descriptor kernels deal a frame's points to the blocks of their
// (blocks per frame, frames) grid. Both gather from the integral image around their points, and the
// gathers miss the L2 a lot when every XCD sees points from all over the frame (64 1080p frames: the
// descriptor kernel fetched 163 MB per frame, twenty times the image). The list is ordered
// (non_max_supression launch, row, column): a launch is one scale, and within a launch the points come in
// raster order. So every XCD takes the same eighth -- a band of image rows -- of EVERY launch
// (consecutive blocks of a grid go to consecutive XCDs; the grid's x extent is a multiple of 8), and an
// XCD's blocks walk that band together, consecutive points to consecutive blocks. Every XCD thus gets
// every scale (one contiguous eighth of the whole list per XCD left the XCD with the coarse points,
// whose windows are far larger, working long after the others: 6.19 instead of 3.39 ms), and its L2
// holds one band of the image.
static_assert(kNms == 6, "PointWalk names the launches' first points one by one");
struct PointWalk {
    // first point of every launch in the frame's list and the end of the list: separate members, not an
    // array (an array indexed by the launch would live in scratch memory); all read up front, side by side
    int s0, s1, s2, s3, s4, s5, s6;
    int xcd, first, stride;          // this block: XCD, position among the XCD's blocks, their number
};

__device__ __forceinline__ PointWalk point_walk(const NmsTable& nms, const int* __restrict__ rowoff, int n)
{
    PointWalk W;
    auto start = [&](int i) { return nms.n[i].row0 < nms.rows ? min(rowoff[nms.n[i].row0], n) : n; };
    W.s0 = start(0); W.s1 = start(1); W.s2 = start(2); W.s3 = start(3); W.s4 = start(4); W.s5 = start(5);
    W.s6 = n;
    W.xcd = (int)(blockIdx.x & 7u);
    W.first = (int)(blockIdx.x >> 3);
    W.stride = (int)(gridDim.x >> 3);
    return W;
}
// True when the block cannot get a point of a list of n points, judged from n alone (a launch holds at
// most n points, an XCD's share at most ceil(n / 8)): the grid is sized for the list's capacity, so for a
// frame with few points most blocks leave here, before anything else is read.
__device__ __forceinline__ bool walk_idle(int n)
{
    const int stride = (int)(gridDim.x >> 3), first = (int)(blockIdx.x >> 3);
    const int most = (n + 7) >> 3;
    bool idle = true;
#pragma unroll
    for (int i = 0; i < kNms; i++) {
        int j = first - i * (stride / kNms);
        if (j < 0) j += stride;
        idle = idle && j >= most;
    }
    return idle;
}
__device__ __forceinline__ int launch_start(const PointWalk& W, int i)
{
    return i == 0 ? W.s0 : i == 1 ? W.s1 : i == 2 ? W.s2 : i == 3 ? W.s3 : i == 4 ? W.s4 : i == 5 ? W.s5 : W.s6;
}
// points of launch i this block handles: lo, lo + stride, ... below hi
__device__ __forceinline__ void walk_range(const PointWalk& W, int i, int* lo, int* hi)
{
    const int s0 = launch_start(W, i);
    const int len = launch_start(W, i + 1) - s0;
    // the launches start at different blocks of the XCD, so that a frame with few points (fewer per
    // launch and XCD than the XCD has blocks) still spreads them over many blocks instead of giving the
    // first blocks one point of every launch
    int j = W.first - i * (W.stride / kNms);
    if (j < 0) j += W.stride;
    *lo = s0 + (int)((long long)W.xcd * len / 8) + j;
    *hi = s0 + (int)((long long)(W.xcd + 1) * len / 8);
}

// atan of a non-negative (possibly infinite or NaN) argument
__device__ __forceinline__ float atan_pos(float v)
{
    if (v != v) return v;
    if (v > 1.0f) return MIF_PIO2_F - mif_atan01(1.0f / v);
    return mif_atan01(v);
}

// getOrientation_kernels.cl:138-153 getAngle
__device__ __forceinline__ float get_angle(float X, float Y)
{
    if (X >= 0 && Y >= 0) return atan_pos(Y / X);
    if (X < 0 && Y >= 0) return kPiF - atan_pos(-Y / X);
    if (X < 0 && Y < 0) return kPiF + atan_pos(Y / X);
    if (X >= 0 && Y < 0) return 2 * kPiF - atan_pos(-Y / X);
    return 0;
}

// getOrientation_kernels.cl:68-71 / createDescriptors_kernel.cl:154-157
__device__ __forceinline__ float gaussian(float x, float y, float sig)
{
    return (1.0f / (2.0f * kPiF * sig * sig)) * mif_expf(-(x * x + y * y) / (2.0f * sig * sig));
}

// getOrientationStep1 + getOrientationStep2 (getOrientation_kernels.cl:161-316)
// in one block of ONE wave per interest point: the lanes evaluate the 109 Haar samples in two turns,
// lanes 0..41 sum the samples of their pi/3 window in sample order, then the reference's max-magnitude
// tree picks the winner. (With two waves per point the second had work in the first phase only; the
// phases of a point follow one another, so what counts is how many points a compute unit has in flight:
// 64 1080p frames 0.53 -> 0.40 ms.)
constexpr int kOriThreads = 64;
__global__ __launch_bounds__(kOriThreads) void k_orientation(const float* __restrict__ img, int w, int h,
                                                     const float2* __restrict__ pos, const float* __restrict__ scale,
                                                     const int* __restrict__ counts, float* __restrict__ orientation,
                                                     NmsTable nms, const int* __restrict__ rowoff)
{
    __shared__ float s_rx[109], s_ry[109], s_rz[109];
    __shared__ float s_sx[42], s_sy[42];
    {
        const size_t f = blockIdx.y;
        img += f * w * h;
        pos += f * kMaxIpts;
        scale += f * kMaxIpts;
        orientation += f * kMaxIpts;
        counts += f * 2;
        rowoff += f * nms.rows;
    }
    const int tid = threadIdx.x;
    const int n_ipts = counts[0];
    // which points the block handles: see PointWalk (uniform over the block)
    if (walk_idle(n_ipts)) return;
    const PointWalk walk = point_walk(nms, rowoff, n_ipts);
    for (int li = 0; li < kNms; li++) {
    int ip_lo, ip_hi;
    walk_range(walk, li, &ip_lo, &ip_hi);
    for (int ip = ip_lo; ip < ip_hi; ip += walk.stride) {
    const int s = (int)mif_roundf(scale[ip]);
    const int r = (int)mif_roundf(pos[ip].y);
    const int c = (int)mif_roundf(pos[ip].x);
    for (int item = tid; item < 109; item += kOriThreads) {
        const int slot = c_orient_slot[item];
        const int localId = c_orient_item[slot];
        const int i = localId / 13 - 6;
        const int j = localId % 13 - 6;
        const float gauss = c_gauss25[7 * c_tab_id[i + 6] + c_tab_id[j + 6]];
        float hx, hy;
        haar_xy(img, w, h, r + j * s, c + i * s, 4 * s, &hx, &hy);
        const float rx = gauss * hx;
        const float ry = gauss * hy;
        s_rx[slot] = rx;
        s_ry[slot] = ry;
        const float ang = get_angle(rx, ry);
        s_rz[slot] = (ang > 0 && ang < 2 * kPiF) ? ang : __builtin_nanf("");     // see the window test below
    }
    __syncthreads();
    // The 42 windows. 0..34 lie inside [0, 2 pi): ang1 < ang && ang < ang2. 35..41 wrap around:
    // (ang > 0 && ang < ang2) || (ang > ang1 && ang < 2 pi). The angles are 0 .. 2 pi or NaN, and the ends
    // of that range (like NaN) lie in no window of either kind, so they are stored as NaN and the wrapped
    // test becomes ang < ang2 || ang > ang1: both kinds share the two comparisons of a sample and differ
    // in how the lane masks combine.
    if (tid < 42) {
        const float ang1 = 0.15f * (float)tid;
        const float ang2 = (ang1 + (kPiF / 3.0f) > 2 * kPiF ? ang1 - (5.0f * kPiF / 3.0f) : ang1 + (kPiF / 3.0f));
        // A sample outside the window contributes s_rx[k] * 0.0f = +-0 in the reference, which leaves the sum
        // as it is (the sum starts at +0 and x + -x is +0, so it never is -0): what is added here is s_rx[k]
        // or 0.0f, chosen by the lane mask that the two comparisons and the kind of window give (scalar
        // operations on the comparisons' masks, then one select per component).
        const uint64_t wrapped = __builtin_amdgcn_ballot_w64(tid > 34);
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int k = 0; k < 109; ++k) {
            const float ang = s_rz[k];
            const uint64_t above = __builtin_amdgcn_ballot_w64(ang > ang1);
            const uint64_t below = __builtin_amdgcn_ballot_w64(ang < ang2);
            const uint64_t in = (above & below) | (wrapped & (above | below));
            float tx, ty;
            asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(tx) : "v"(s_rx[k]), "s"(in));
            asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(ty) : "v"(s_ry[k]), "s"(in));
            sx += tx;
            sy += ty;
        }
        s_sx[tid] = sx;
        s_sy[tid] = sy;
    }
    __syncthreads();
    for (int offset = 32; offset > 0; offset >>= 1) {
        if (tid < offset && tid + offset < 42) {
            const float m0 = s_sx[tid] * s_sx[tid] + s_sy[tid] * s_sy[tid];
            const float m1 = s_sx[tid + offset] * s_sx[tid + offset] + s_sy[tid + offset] * s_sy[tid + offset];
            if (m0 < m1) {
                s_sx[tid] = s_sx[tid + offset];
                s_sy[tid] = s_sy[tid + offset];
            }
        }
        __syncthreads();
    }
    if (tid == 0) orientation[ip] = get_angle(s_sx[0], s_sy[0]);
    __syncthreads();                                 // the shared arrays are reused by the next point
    }
    }
}

// -------------------------------------------------------------- descriptor

constexpr int kIpointWords = (int)(sizeof(mif_ipoint) / 4);
static_assert(kIpointWords == 70, "mif_ipoint is 70 words");

// createDescriptors (createDescriptors_kernel.cl:160-275) fused with
// normalizeDescriptors (normalizeDescriptors_kernel.cl:54-111): one block of
// 256 threads per interest point evaluates the 16 x 81 rotated Haar samples,
// 64 threads then run the reference's reduction tree (81 -> 64 -> 32 ... -> 1)
// per sub-region and component, and the 16 partial squared lengths are summed
// with its 8-4-2-1 tree.
__global__ __launch_bounds__(256) void k_descriptor(const float* __restrict__ img, int width, int height,
                                                    const float2* __restrict__ pos, const float* __restrict__ scale,
                                                    const float* __restrict__ orientation,
                                                    const int* __restrict__ counts, float* __restrict__ desc,
                                                    NmsTable nms, const int* __restrict__ rowoff,
                                                    const int* __restrict__ lap, uint32_t* __restrict__ packed)
{
    constexpr int kGridN = 24;                           // sample positions per axis: k, l = -12 .. 11
    __shared__ float s_d[16][4][81];
    __shared__ float s_g[4][kGridN * kGridN];            // per position: Haar x, Haar y, image column, image row
    __shared__ float s_len[16];
    {
        const size_t f = blockIdx.y;
        img += f * width * height;
        pos += f * kMaxIpts;
        scale += f * kMaxIpts;
        orientation += f * kMaxIpts;
        desc += f * kMaxIpts * 64;
        counts += f * 2;
        rowoff += f * nms.rows;
        lap += f * kMaxIpts;
        packed += f * kMaxIpts * kIpointWords;
    }
    const int tid = threadIdx.x;
    const int n_ipts = counts[0];
    // which points the block handles: see PointWalk (uniform over the block)
    if (walk_idle(n_ipts)) return;
    const PointWalk walk = point_walk(nms, rowoff, n_ipts);
    for (int li = 0; li < kNms; li++) {
    int ip_lo, ip_hi;
    walk_range(walk, li, &ip_lo, &ip_hi);
    for (int ip = ip_lo; ip < ip_hi; ip += walk.stride) {
    const float x = mif_roundf(pos[ip].x);
    const float y = mif_roundf(pos[ip].y);
    float si, co;
    mif_sincosf(orientation[ip], &si, &co);
    const float thScale = scale[ip];
    const int hs = (int)(2 * mif_roundf(thScale));
    // The 81 samples of a sub-region form a 9 x 9 grid (k, l) rotated by the orientation: k moves a
    // sample by thScale (co, si) pixels, l by thScale (-si, co). The grids of neighbouring sub-regions
    // start 5 steps apart and so share 4 of their 9 rows and columns: the 16 x 81 samples lie on 24 x 24
    // positions (k, l = -12 .. 11), and a sample's position and Haar responses depend on (k, l) alone
    // (createDescriptors_kernel.cl:225-236), only its Gaussian weight on the sub-region. The responses are
    // therefore evaluated once per position, 576 instead of 1296 times (eight reads of the integral
    // image each, on different image rows from lane to lane: their number is what bounds the kernel), and
    // every sample then takes its position's values from LDS.
    // Consecutive threads take consecutive positions along the axis that changes the image COLUMN more,
    // so that the lanes of a wave read along image rows rather than down a column (a cache line per lane).
#ifdef MIF_SURF_NO_KFAST
    const bool k_fast = false;
#else
    const bool k_fast = mif_fabsf(co) >= mif_fabsf(si);
#endif
    for (int item = tid; item < kGridN * kGridN; item += 256) {
        const int a = item / kGridN, b = item - a * kGridN;
        const int kk = k_fast ? b : a, ll = k_fast ? a : b;
        const int k = kk - 12, l = ll - 12;
        const int sample_x = (int)mif_roundf(x + ((float)(-l) * thScale * si + (float)k * thScale * co));
        const int sample_y = (int)mif_roundf(y + ((float)l * thScale * co + (float)k * thScale * si));
        float rx, ry;
        if (hs == 4 || hs == 6) haar_xy_small(img, width, height, sample_y, sample_x, hs, &rx, &ry);    // uniform
        else haar_xy(img, width, height, sample_y, sample_x, hs, &rx, &ry);
        const int slot = kk * kGridN + ll;
        s_g[0][slot] = rx;
        s_g[1][slot] = ry;
        s_g[2][slot] = (float)sample_x;
        s_g[3][slot] = (float)sample_y;
    }
    __syncthreads();
    for (int item = tid; item < 16 * 81; item += 256) {
        const int bIdx = item / 81;
        const int tha = item - bIdx * 81;
        const int j = c_tab_j[bIdx];
        const int i = c_tab_i[bIdx];
        const float ix = (float)(i + 5);
        const float jx = (float)(j + 5);
        const float xs = mif_roundf(x + (-jx * thScale * si + ix * thScale * co));
        const float ys = mif_roundf(y + (jx * thScale * co + ix * thScale * si));
        const int k = i + (tha / 9);
        const int l = j + (tha % 9);
        const int slot = (k + 12) * kGridN + (l + 12);
        const float gauss_s1 = gaussian(xs - s_g[2][slot], ys - s_g[3][slot], 2.5f * thScale);
        const float rx = s_g[0][slot], ry = s_g[1][slot];
        const float rrx = gauss_s1 * (-rx * si + ry * co);
        const float rry = gauss_s1 * (rx * co + ry * si);
        s_d[bIdx][0][tha] = rrx;
        s_d[bIdx][1][tha] = rry;
        s_d[bIdx][2][tha] = mif_fabsf(rrx);
        s_d[bIdx][3][tha] = mif_fabsf(rry);
    }
    __syncthreads();
    // sumDesc (createDescriptors_kernel.cl:133-150) over the 64 arrays of 81 values at once: the step
    // 81 -> 64 and then strides 32 .. 1, the additions of a step spread over all threads
    {
        float* flat = &s_d[0][0][0];
        for (int item = tid; item < 64 * (81 - 64); item += 256) {
            const int arr = item / (81 - 64), t = item - arr * (81 - 64);
            flat[arr * 81 + t] += flat[arr * 81 + t + 64];
        }
        __syncthreads();
        for (int stride = 32; stride > 0; stride >>= 1) {
            for (int item = tid; item < 64 * stride; item += 256) {
                const int arr = item / stride, t = item - arr * stride;
                flat[arr * 81 + t] += flat[arr * 81 + t + stride];
            }
            __syncthreads();
        }
    }
    float mine = 0.0f;
    if (tid < 64) {
        const int bIdx = tid >> 2, q = tid & 3;
        float* v = s_d[bIdx][q];
        float cx = 0.5f, cy = 0.5f;
        cx += (float)(bIdx / 4);
        cy += (float)(bIdx % 4);
        mine = v[0] * gaussian(cx - 2.0f, cy - 2.0f, 1.5f);
        v[0] = mine;
    }
    __syncthreads();
    if (tid < 16) {
        const float a = s_d[tid][0][0], b = s_d[tid][1][0], c = s_d[tid][2][0], d = s_d[tid][3][0];
        s_len[tid] = a * a + b * b + c * c + d * d;
    }
    __syncthreads();
    for (int i = 8; i > 0; i >>= 1) {
        if (tid < i) s_len[tid] += s_len[tid + i];
        __syncthreads();
    }
    // the point's record (mif_ipoint: x, y, scale, orientation, laplacian, 0, descriptor[64]; the layout of
    // ccl_surf::Ipoint, ccl_surf.h:11-21) is written here as well, so that a host caller fetches the points
    // of a frame with one copy
    if (tid < 64) {
        const float lengthOfDescriptor = 1.0f / __builtin_sqrtf(s_len[0]);
        const float d = mine * lengthOfDescriptor;
        desc[(size_t)ip * 64 + tid] = d;
        packed[(size_t)ip * kIpointWords + 6 + tid] = __float_as_uint(d);
    } else if (tid < 64 + 6) {
        const int k = tid - 64;
        const uint32_t v = k == 0 ? __float_as_uint(pos[ip].x) : k == 1 ? __float_as_uint(pos[ip].y)
                         : k == 2 ? __float_as_uint(scale[ip]) : k == 3 ? __float_as_uint(orientation[ip])
                         : k == 4 ? (uint32_t)lap[ip] : 0u;
        packed[(size_t)ip * kIpointWords + k] = v;
    }
    __syncthreads();                                 // the shared arrays are reused by the next point
    }
    }
}

} // namespace


struct mif_surf {
    int device = 0, W = 0, H = 0, C = 1, batch = 1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // optional per-stage timing: events at the stage boundaries of a batch
    bool timing = false, timed = false;
    hipEvent_t ev_stage[MIF_SURF_STAGES + 1] = {};
    LayerTable tab;
    NmsTable nms;
    size_t pool = 0;
    int nms_max_h = 0;
    uint8_t* d_pixels = nullptr;
    float* integral = nullptr;
    // every buffer holds `batch` frames back to back
    // responses | laplacians | mask | rowcnt live in ONE allocation that a single memset clears per batch
    void* cleared = nullptr;
    size_t cleared_bytes = 0;
    float* responses = nullptr;
    int* laplacians = nullptr;
    unsigned long long* mask = nullptr;
    int* rowcnt = nullptr;
    unsigned int* ticket = nullptr;      // k_integral_fused: band numbers; cleared with the rest
    int* fault = nullptr;                // k_integral_fused: a band gave up waiting
    unsigned long long* carry = nullptr; // k_integral_fused: [batch][bands][columns padded to 64] {total, launch}
    unsigned int epoch = 0;              // launch number of the last fused launch
    int* fallbacks = nullptr;            // batches that the launch-per-pass form had to redo (device counter)
    mif_ipoint* packed = nullptr;        // the points as mif_ipoint records, for a single download
    int* rowoff = nullptr;
    int* counts = nullptr;               // {points (cut at 8000), raw total}
    float2* pos = nullptr;
    float* scale = nullptr;
    int* lap = nullptr;
    float* orient = nullptr;
    float* desc = nullptr;
    int* h_counts = nullptr;
};

namespace {

void surf_free(mif_surf* s)
{
    (void)hipFree(s->d_pixels); (void)hipFree(s->integral); (void)hipFree(s->cleared); (void)hipFree(s->rowoff);
    (void)hipFree(s->packed); (void)hipFree(s->carry); (void)hipFree(s->fallbacks);
    (void)hipFree(s->counts); (void)hipFree(s->pos); (void)hipFree(s->scale); (void)hipFree(s->lap);
    (void)hipFree(s->orient); (void)hipFree(s->desc);
    if (s->h_counts) (void)hipHostFree(s->h_counts);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    for (hipEvent_t e : s->ev_stage)
        if (e) (void)hipEventDestroy(e);
    if (s->stream) (void)hipStreamDestroy(s->stream);
}

int surf_create_impl(mif_surf* s)
{
    MIF_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    MIF_HIP(hipEventCreate(&s->ev0));
    MIF_HIP(hipEventCreate(&s->ev1));
    // ccl_surf.cpp:115-153 createResponseMap for OCTAVES = 3, SAMPLE_STEP = 5
    const int w = s->W / kSampleStep, h = s->H / kSampleStep, st = kSampleStep;
    const int spec[kLayers][4] = {{w, h, st, 9}, {w, h, st, 15}, {w, h, st, 21}, {w, h, st, 27},
                                  {w / 2, h / 2, st * 2, 39}, {w / 2, h / 2, st * 2, 51},
                                  {w / 4, h / 4, st * 4, 75}, {w / 4, h / 4, st * 4, 99}};
    size_t off = 0;
    for (int i = 0; i < kLayers; i++) {
        s->tab.l[i].w = spec[i][0]; s->tab.l[i].h = spec[i][1];
        s->tab.l[i].step = spec[i][2]; s->tab.l[i].filter = spec[i][3];
        s->tab.l[i].off = off;
        off += (size_t)spec[i][0] * spec[i][1];
    }
    s->pool = off > 0 ? off : 1;
    // ccl_surf.cpp:109-113 filter_map, 302-337 launch order
    const int filter_map[kOctaves][4] = {{0, 1, 2, 3}, {1, 3, 4, 5}, {3, 5, 6, 7}};
    int rows = 0, max_w = 1, max_h = 1, n = 0;
    for (int o = 0; o < kOctaves; o++)
        for (int i = 0; i <= 1; i++) {
            NmsDesc& d = s->nms.n[n++];
            d.b = filter_map[o][i]; d.m = filter_map[o][i + 1]; d.t = filter_map[o][i + 2];
            d.row0 = rows;
            rows += s->tab.l[d.m].h;
            max_w = std::max(max_w, s->tab.l[d.m].w);
            max_h = std::max(max_h, s->tab.l[d.m].h);
        }
    s->nms.rows = rows;
    s->nms.words = (max_w + 63) / 64;
    s->nms_max_h = max_h;
    const size_t px = (size_t)s->W * s->H;
    const size_t B = (size_t)s->batch;
    MIF_HIP(hipMalloc(&s->d_pixels, px * s->C));
    MIF_HIP(hipMalloc(&s->integral, B * px * sizeof(float)));
    {
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t b_resp = up(B * s->pool * sizeof(float)), b_lap = up(B * s->pool * sizeof(int));
        const size_t b_mask = up(B * (size_t)std::max(rows, 1) * s->nms.words * sizeof(unsigned long long));
        const size_t b_cnt = up(B * (size_t)std::max(rows, 1) * sizeof(int));
        s->cleared_bytes = b_resp + b_lap + b_mask + b_cnt + 256;
        MIF_HIP(hipMalloc(&s->cleared, s->cleared_bytes));
        char* base = (char*)s->cleared;
        s->responses = (float*)base;
        s->laplacians = (int*)(base + b_resp);
        s->mask = (unsigned long long*)(base + b_resp + b_lap);
        s->rowcnt = (int*)(base + b_resp + b_lap + b_mask);
        s->ticket = (unsigned int*)(base + b_resp + b_lap + b_mask + b_cnt);
        s->fault = (int*)(s->ticket + 1);
        const size_t carry_bytes = B * (size_t)((s->H + 63) / 64) * (size_t)((s->W + 63) / 64 * 64) * sizeof(unsigned long long);
        MIF_HIP(hipMalloc(&s->carry, carry_bytes));
        MIF_HIP(hipMemset(s->carry, 0, carry_bytes));
        MIF_HIP(hipMalloc(&s->fallbacks, sizeof(int)));
        MIF_HIP(hipMemset(s->fallbacks, 0, sizeof(int)));
    }
    MIF_HIP(hipMalloc(&s->packed, B * kMaxIpts * sizeof(mif_ipoint)));
    MIF_HIP(hipMalloc(&s->rowoff, B * (size_t)std::max(rows, 1) * sizeof(int)));
    MIF_HIP(hipMalloc(&s->counts, B * 2 * sizeof(int)));
    MIF_HIP(hipMemset(s->counts, 0, B * 2 * sizeof(int)));
    MIF_HIP(hipMalloc(&s->pos, B * kMaxIpts * sizeof(float2)));
    MIF_HIP(hipMalloc(&s->scale, B * kMaxIpts * sizeof(float)));
    MIF_HIP(hipMalloc(&s->lap, B * kMaxIpts * sizeof(int)));
    MIF_HIP(hipMalloc(&s->orient, B * kMaxIpts * sizeof(float)));
    MIF_HIP(hipMalloc(&s->desc, B * kMaxIpts * 64 * sizeof(float)));
    MIF_HIP(hipHostMalloc(&s->h_counts, B * 2 * sizeof(int)));
    unsigned char items[109];
    int k = 0;
    for (int id = 0; id < 169; id++) {
        const int i = id / 13 - 6, j = id % 13 - 6;
        if (i * i + j * j < 36) items[k++] = (unsigned char)id;
    }
    MIF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_orient_item), items, sizeof(items)));
    unsigned char slots[109];
    for (int t = 0; t < 109; t++) slots[t] = (unsigned char)t;
    std::sort(slots, slots + 109, [&](unsigned char a, unsigned char b) {
        const int ja = items[a] % 13, jb = items[b] % 13;          // row of the grid
        return ja != jb ? ja < jb : items[a] < items[b];           // then column (id / 13)
    });
    MIF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_orient_slot), slots, sizeof(slots)));
    return MIF_OK;
}

// Enqueues the whole pipeline for `nf` frames (u8 [nf][H][W][C], back to back in device memory): every
// launch covers all frames of the batch.
int surf_enqueue(mif_surf* s, const void* d_pixels, int nf)
{
    hipStream_t st = s->stream;
    const int w = s->W, h = s->H;
    MIF_HIP(hipEventRecord(s->ev0, st));
    const bool tm = s->timing;
    s->timed = false;
    if (tm) MIF_HIP(hipEventRecord(s->ev_stage[0], st));
    // layers that are empty keep zero-sized slots; the pools are cleared so that
    // positions no work item writes read as zero, as in freshly allocated buffers
    // (one memset: response and laplacian pools, acceptance mask, row counts)
    MIF_HIP(hipMemsetAsync(s->cleared, 0, s->cleared_bytes, st));
    // both passes of the integral image in one launch for batches (a single frame has too few bands to fill
    // the device: the two-pass form is faster there); MIF_SURF_INTEGRAL=fused / twopass forces a form
    static const char* form_env = getenv("MIF_SURF_INTEGRAL");
    const int nbands = (h + 63) / 64;
    const bool fused = w * h >= 2 && (form_env ? strcmp(form_env, "fused") == 0 : nbands * nf >= kFusedMinBands);
    if (fused) {
        if (++s->epoch == 0) {                              // the launch numbers have wrapped: forget the old words
            MIF_HIP(hipMemsetAsync(s->carry, 0, (size_t)s->batch * nbands * ((w + 63) / 64 * 64) * sizeof(unsigned long long), st));
            s->epoch = 1;
        }
        // (four pixels per load where every row of every frame starts on a 4-byte boundary)
        const bool a4 = w % 4 == 0 && (uintptr_t)d_pixels % 4 == 0;
        // (MIF_SURF_INTEGRAL_TEST_FAULT: the bands wait for a launch number nobody writes, give up, and the
        // launches behind redo the batch; for the test of that path)
        const char* test_fault = getenv("MIF_SURF_INTEGRAL_TEST_FAULT");
        const unsigned int epoch_wait = (test_fault && test_fault[0] == '1') ? s->epoch + 0x40000000u : s->epoch;
#define MIF_LAUNCH_FUSED(CH, A4)                                                                                          \
    hipLaunchKernelGGL((k_integral_fused<CH, A4>), dim3(nbands * nf), dim3(256), 0, st, (const uint8_t*)d_pixels,          \
                       s->integral, w, h, nbands, nbands * nf, s->carry, (w + 63) / 64 * 64, s->epoch, epoch_wait,         \
                       s->ticket, s->fault)
        if (s->C == 3) { if (a4) MIF_LAUNCH_FUSED(3, true); else MIF_LAUNCH_FUSED(3, false); }
        else { if (a4) MIF_LAUNCH_FUSED(1, true); else MIF_LAUNCH_FUSED(1, false); }
#undef MIF_LAUNCH_FUSED
    }
    // a launch per pass: the form of single frames and small batches, and behind the one-launch form the
    // stand-in that runs only for a batch whose bands gave up waiting for each other
    {
        const int* only_if = fused ? s->fault : nullptr;
        hipLaunchKernelGGL(k_scan_rows, dim3((h + 3) / 4, nf), dim3(256), 0, st, (const uint8_t*)d_pixels, s->integral, w, h,
                           s->C, only_if);
        if ((h + 63) / 64 <= kColMaxBlocks)
            hipLaunchKernelGGL(k_scan_cols, dim3((w + 63) / 64, nf), dim3(64 * kColWaves), 0, st, s->integral, w, h, only_if);
        else
            hipLaunchKernelGGL(k_scan_cols_serial, dim3((w + 63) / 64, nf), dim3(64), 0, st, s->integral, w, h, only_if);
        if (fused) hipLaunchKernelGGL(k_fault_handled, dim3(1), dim3(1), 0, st, s->fault, s->fallbacks);
    }
    if (tm) MIF_HIP(hipEventRecord(s->ev_stage[1], st));
    // the first octave's four layers from a window staged in LDS where they share one output grid (they do
    // for every image the reference's layer table is built for), the rest straight from memory
    static const bool no_staged = getenv("MIF_SURF_NO_STAGED_HESSIAN") != nullptr;      // A/B switch
    bool staged = !no_staged && s->tab.l[0].w > 0 && s->tab.l[0].h > 0;
    for (int i = 0; i < kHessFirstLayers; i++)
        staged = staged && s->tab.l[i].w == s->tab.l[0].w && s->tab.l[i].h == s->tab.l[0].h &&
                 s->tab.l[i].step == kSampleStep && s->tab.l[i].filter <= kHessMaxFilter;
    const int first = staged ? kHessFirstLayers : 0;
    if (staged)
        hipLaunchKernelGGL(k_hessian_staged, dim3((s->tab.l[0].w + 15) / 16, (s->tab.l[0].h + 15) / 16, nf), dim3(256), 0,
                           st, (const float*)s->integral, w, h, s->tab, s->responses, s->laplacians, s->pool);
    const int lw = std::max(s->tab.l[first].w, 1), lh = std::max(s->tab.l[first].h, 1);
    hipLaunchKernelGGL(k_hessian, dim3((lw + 15) / 16, (lh + 15) / 16, (kLayers - first) * nf), dim3(256), 0, st,
                       (const float*)s->integral, w, h, s->tab, s->responses, s->laplacians, s->pool, first);
    if (tm) MIF_HIP(hipEventRecord(s->ev_stage[2], st));
    if (s->nms.rows > 0)
        hipLaunchKernelGGL(k_nms_mark, dim3(s->nms.words, (s->nms.rows + 3) / 4, nf), dim3(256), 0, st, s->tab, s->nms,
                       (const float*)s->responses, (const int*)s->laplacians, kThres, s->mask, s->rowcnt, s->pool);
    hipLaunchKernelGGL(k_scan_counts, dim3(nf), dim3(1024), 0, st, (const int*)s->rowcnt, s->rowoff, s->nms.rows,
                       kMaxIpts, s->counts, (const int*)s->fault);
    if (s->nms.rows > 0)
        hipLaunchKernelGGL(k_nms_emit, dim3((s->nms.rows + 3) / 4, nf), dim3(256), 0, st, s->tab, s->nms,
                           (const float*)s->responses, (const int*)s->laplacians,
                           (const unsigned long long*)s->mask, (const int*)s->rowcnt, (const int*)s->rowoff, kMaxIpts,
                           s->pos, s->scale, s->lap, s->pool);
    // The point lists hold up to 8000 points per frame and a typical frame a few thousand: the blocks of a
    // frame walk its list (see PointWalk). One frame gets a block per two list slots, batches get fewer
    // blocks per frame (64 frames: 1024, measured against 256 / 2048 / 4000: descriptor 3.00 ms against
    // 3.16 / 3.15 / 3.42; few blocks per frame keep many frames in flight, whose bands of the integral
    // image then compete for the L2).
    static const int per_frame_env = getenv("MIF_SURF_BLOCKS") ? atoi(getenv("MIF_SURF_BLOCKS")) : 0;   // tuning override
    const int per_frame = per_frame_env > 0 ? 8 * ((per_frame_env + 7) / 8)
                                            : 8 * ((std::max(64, std::min(kMaxIpts / 2, 65536 / nf)) + 7) / 8);
    if (tm) MIF_HIP(hipEventRecord(s->ev_stage[3], st));
    hipLaunchKernelGGL(k_orientation, dim3(per_frame, nf), dim3(kOriThreads), 0, st, (const float*)s->integral, w, h,
                       (const float2*)s->pos, (const float*)s->scale, (const int*)s->counts, s->orient, s->nms,
                       (const int*)s->rowoff);
    if (tm) MIF_HIP(hipEventRecord(s->ev_stage[4], st));
    hipLaunchKernelGGL(k_descriptor, dim3(per_frame, nf), dim3(256), 0, st, (const float*)s->integral, w, h,
                       (const float2*)s->pos, (const float*)s->scale, (const float*)s->orient, (const int*)s->counts,
                       s->desc, s->nms, (const int*)s->rowoff, (const int*)s->lap, reinterpret_cast<uint32_t*>(s->packed));
    MIF_HIP(hipGetLastError());
    if (tm) {
        MIF_HIP(hipEventRecord(s->ev_stage[5], st));
        s->timed = true;
    }
    MIF_HIP(hipEventRecord(s->ev1, st));
    return MIF_OK;
}

} // namespace

extern "C" {

int mif_surf_create(int device, int width, int height, int channels, mif_surf** out)
{
    return mif_surf_create_batched(device, width, height, channels, 1, out);
}

int mif_surf_batch(const mif_surf* s) { return s ? s->batch : MIF_ERR_INVALID_ARGUMENT; }

int mif_surf_create_batched(int device, int width, int height, int channels, int batch, mif_surf** out)
{
    if (!out) { mif::set_error("mif_surf_create: out is NULL"); return MIF_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    if (width <= 0 || height <= 0 || (channels != 1 && channels != 3) ||
        (long long)width * height * 3 >= (1ll << 31) || batch < 1 || batch > 8190) {
        mif::set_error("mif_surf_create: bad geometry %dx%dx%d, batch %d", width, height, channels, batch);
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(device);
    if (!guard.ok) { mif::set_error("mif_surf_create: cannot select device %d", device); return MIF_ERR_HIP; }
    mif_surf* s = new mif_surf();
    s->device = device; s->W = width; s->H = height; s->C = channels; s->batch = batch;
    const int rc = surf_create_impl(s);
    if (rc != MIF_OK) { surf_free(s); delete s; return rc; }
    *out = s;
    return MIF_OK;
}

int mif_surf_destroy(mif_surf* s)
{
    if (!s) return MIF_OK;
    mif::DeviceGuard guard(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    surf_free(s);
    delete s;
    return MIF_OK;
}

int mif_surf_capacity(const mif_surf* s) { return s ? kMaxIpts : MIF_ERR_INVALID_ARGUMENT; }

int mif_surf_enqueue(mif_surf* s, const void* d_pixels)
{
    if (!s || !d_pixels) { mif::set_error("mif_surf_enqueue: NULL argument"); return MIF_ERR_INVALID_ARGUMENT; }
    mif::DeviceGuard guard(s->device);
    return surf_enqueue(s, d_pixels, 1);
}

int mif_surf_enqueue_batch(mif_surf* s, const void* d_pixels, int nframes)
{
    if (!s || !d_pixels) { mif::set_error("mif_surf_enqueue_batch: NULL argument"); return MIF_ERR_INVALID_ARGUMENT; }
    if (nframes < 1 || nframes > s->batch) {
        mif::set_error("mif_surf_enqueue_batch: %d frames do not fit the handle's batch of %d", nframes, s->batch);
        return MIF_ERR_CAPACITY;
    }
    mif::DeviceGuard guard(s->device);
    return surf_enqueue(s, d_pixels, nframes);
}

int mif_surf_copy_points(mif_surf* s, int frame, mif_ipoint* points, int capacity, int* count, int* raw_count)
{
    if (!s || !count || (capacity > 0 && !points) || frame < 0 || frame >= s->batch) {
        mif::set_error("mif_surf_copy_points: bad argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(s->device);
    *count = 0;
    MIF_HIP(hipMemcpyAsync(s->h_counts + 2 * frame, s->counts + 2 * frame, 2 * sizeof(int), hipMemcpyDeviceToHost,
                           s->stream));
    MIF_HIP(hipStreamSynchronize(s->stream));
    const int n = s->h_counts[2 * frame];
    if (s->h_counts[2 * frame + 1] == -2) {
        mif::set_error("mif_surf_copy_points: the integral image pass of this batch did not complete");
        return MIF_ERR_HIP;
    }
    if (raw_count) *raw_count = s->h_counts[2 * frame + 1];
    if (n > capacity) {
        mif::set_error("mif_surf_copy_points: %d interest points do not fit capacity %d", n, capacity);
        return MIF_ERR_CAPACITY;
    }
    if (n > 0)
        MIF_HIP(hipMemcpy(points, s->packed + (size_t)frame * kMaxIpts, (size_t)n * sizeof(mif_ipoint),
                          hipMemcpyDeviceToHost));
    *count = n;
    return MIF_OK;
}

int mif_surf_sync(mif_surf* s)
{
    if (!s) return MIF_ERR_INVALID_ARGUMENT;
    mif::DeviceGuard guard(s->device);
    MIF_HIP(hipStreamSynchronize(s->stream));
    return MIF_OK;
}

void* mif_surf_stream(mif_surf* s) { return s ? (void*)s->stream : nullptr; }

int mif_surf_device_results(mif_surf* s, void** d_pos, void** d_scale, void** d_laplacian, void** d_orientation,
                            void** d_descriptors, void** d_counts)
{
    if (!s) return MIF_ERR_INVALID_ARGUMENT;
    if (d_pos) *d_pos = s->pos;
    if (d_scale) *d_scale = s->scale;
    if (d_laplacian) *d_laplacian = s->lap;
    if (d_orientation) *d_orientation = s->orient;
    if (d_descriptors) *d_descriptors = s->desc;
    if (d_counts) *d_counts = s->counts;
    return MIF_OK;
}

int mif_surf_run(mif_surf* s, const uint8_t* pixels, mif_ipoint* points, int capacity, int* count, int* raw_count,
                 float* elapsed_ms)
{
    if (!s || !pixels || !count || (capacity > 0 && !points)) {
        mif::set_error("mif_surf_run: NULL argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(s->device);
    *count = 0;
    MIF_HIP(hipMemcpyAsync(s->d_pixels, pixels, (size_t)s->W * s->H * s->C, hipMemcpyHostToDevice, s->stream));
    const int rc = surf_enqueue(s, s->d_pixels, 1);
    if (rc != MIF_OK) return rc;
    MIF_HIP(hipMemcpyAsync(s->h_counts, s->counts, 2 * sizeof(int), hipMemcpyDeviceToHost, s->stream));
    MIF_HIP(hipStreamSynchronize(s->stream));
    if (elapsed_ms) MIF_HIP(hipEventElapsedTime(elapsed_ms, s->ev0, s->ev1));
    const int n = s->h_counts[0];
    if (s->h_counts[1] == -2) {
        mif::set_error("mif_surf_run: the integral image pass did not complete");
        return MIF_ERR_HIP;
    }
    if (raw_count) *raw_count = s->h_counts[1];
    if (n > capacity) {
        mif::set_error("mif_surf_run: %d interest points do not fit capacity %d", n, capacity);
        return MIF_ERR_CAPACITY;
    }
    if (n == 0) return MIF_OK;
    MIF_HIP(hipMemcpy(points, s->packed, (size_t)n * sizeof(mif_ipoint), hipMemcpyDeviceToHost));
    *count = n;
    return MIF_OK;
}

int mif_surf_integral_fallbacks(mif_surf* s, int* count)
{
    if (!s || !count) return MIF_ERR_INVALID_ARGUMENT;
    mif::DeviceGuard guard(s->device);
    MIF_HIP(hipStreamSynchronize(s->stream));
    MIF_HIP(hipMemcpy(count, s->fallbacks, sizeof(int), hipMemcpyDeviceToHost));
    return MIF_OK;
}

int mif_surf_set_timing(mif_surf* s, int enable)
{
    if (!s) return MIF_ERR_INVALID_ARGUMENT;
    mif::DeviceGuard guard(s->device);
    if (enable && !s->ev_stage[0])
        for (hipEvent_t& e : s->ev_stage) MIF_HIP(hipEventCreate(&e));
    s->timing = enable != 0;
    return MIF_OK;
}

int mif_surf_stage_times(mif_surf* s, float ms[MIF_SURF_STAGES])
{
    if (!s || !ms) return MIF_ERR_INVALID_ARGUMENT;
    if (!s->timed) {
        mif::set_error("mif_surf_stage_times: no batch has been enqueued with timing enabled");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(s->device);
    MIF_HIP(hipStreamSynchronize(s->stream));
    for (int i = 0; i < MIF_SURF_STAGES; i++) MIF_HIP(hipEventElapsedTime(&ms[i], s->ev_stage[i], s->ev_stage[i + 1]));
    return MIF_OK;
}

int mif_surf_copy_integral(mif_surf* s, float* out) { return mif_surf_copy_integral_frame(s, 0, out); }

int mif_surf_copy_integral_frame(mif_surf* s, int frame, float* out)
{
    if (!s || !out || frame < 0 || frame >= s->batch) return MIF_ERR_INVALID_ARGUMENT;
    mif::DeviceGuard guard(s->device);
    MIF_HIP(hipStreamSynchronize(s->stream));
    MIF_HIP(hipMemcpy(out, s->integral + (size_t)frame * s->W * s->H, (size_t)s->W * s->H * sizeof(float),
                      hipMemcpyDeviceToHost));
    return MIF_OK;
}

int mif_surf_layer_info(const mif_surf* s, int layer, int* width, int* height, int* step, int* filter)
{
    if (!s || layer < 0 || layer >= kLayers) return MIF_ERR_INVALID_ARGUMENT;
    if (width) *width = s->tab.l[layer].w;
    if (height) *height = s->tab.l[layer].h;
    if (step) *step = s->tab.l[layer].step;
    if (filter) *filter = s->tab.l[layer].filter;
    return MIF_OK;
}

int mif_surf_copy_layer(mif_surf* s, int layer, float* responses, int32_t* laplacians)
{
    if (!s || layer < 0 || layer >= kLayers) return MIF_ERR_INVALID_ARGUMENT;
    mif::DeviceGuard guard(s->device);
    MIF_HIP(hipStreamSynchronize(s->stream));
    const size_t n = (size_t)s->tab.l[layer].w * s->tab.l[layer].h;
    if (n == 0) return MIF_OK;
    if (responses)
        MIF_HIP(hipMemcpy(responses, s->responses + s->tab.l[layer].off, n * sizeof(float), hipMemcpyDeviceToHost));
    if (laplacians)
        MIF_HIP(hipMemcpy(laplacians, s->laplacians + s->tab.l[layer].off, n * sizeof(int), hipMemcpyDeviceToHost));
    return MIF_OK;
}

} // extern "C"
