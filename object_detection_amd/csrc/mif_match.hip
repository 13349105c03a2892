// Brute-force L2 two-nearest-neighbour descriptor matching on the MI355X matrix
// cores. Replaces the cv::FlannBasedMatcher::match / knnMatch(k=2) calls of the
// reference (common/kpmatch.h:174-176, 284-286) with an exact search.
//
// Definition (fixed so that every run and every machine returns the same bits):
//   nq_i  = sum_k q_ik^2, nd_j = sum_k d_jk^2        fmaf chains, k ascending from 0
//   dot_ij = sum_k q_ik d_jk                          fmaf chain, k ascending from 0
//   s_ij  = fmaf(-2, dot_ij, nq_i + nd_j)             ranking score (squared L2)
//   the two smallest (s_ij, j) in lexicographic order win (ties -> lowest index)
//   reported distance = sqrt(sum_k (q_ik - d_jk)^2)   fmaf chain, k ascending,
//   and the two winners are swapped if that puts the smaller distance first.
// v_mfma_f32_32x32x2_f32 accumulates exactly such a k-ordered fmaf chain, so the
// GEMM below reproduces dot_ij bit for bit.
#include <algorithm>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include "mif_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kQT = 32;      // query rows per wave (one 32x32 MFMA tile)
constexpr int kWaves = 4;    // waves per block, each with its own query tile
constexpr int kDC = 64;      // database rows per staged chunk (two MFMA tiles)
constexpr float kInf = __builtin_inff();
constexpr int kMaxPairs = 256;  // pairs per launch (their item ranges are cached in LDS)
constexpr int kTableSlots = 4;  // pair tables in flight per matcher
constexpr int kCandCap = 32;    // candidates kept per query by the screening passes
// Screening (see k_match_screen): a bf16 score differs from the fp32 ranking score by at most
// kScreenRel * |q| * |d| + kScreenAbs * (|q|^2 + |d|^2). bf16 keeps 8 significant bits, so rounding
// an operand to nearest moves it by at most 2^-8 of its value; a product of two rounded operands is
// off by at most (2 * 2^-8 + 2^-16) |q_k d_k| (the products themselves are exact in the matrix unit);
// the 128 accumulations of the matrix unit are allowed 2^-22 each and the fp32 chain it is compared
// with has 128 roundings of 2^-24 of its own. With sum |q_k d_k| <= |q| |d| and the factor 2 of the
// score: 2 * (2 * 2^-8 + 2^-16 + 128 * 2^-22 + 128 * 2^-24) = 0.01573, rounded up. The absolute term
// covers the roundings of |q|^2 + |d|^2 and of the final fused multiply-add.
constexpr float kScreenRel = 0.016f, kScreenAbs = 1e-6f;

// The pair table of a call lives in device memory (copied there on the call's stream), so that
// one launch can serve a whole batch of frame pairs: few launches mean few tails.
typedef const mif_match_pair* __restrict__ PairTable;

// Loads `n` (<= 32) consecutive floats of a row into registers with as few
// round trips to memory as possible: 16-byte loads when the row allows it.
template <bool VEC>
__device__ __forceinline__ void load_span(const float* __restrict__ r, int n, float (&v)[32])
{
    if (VEC && n == 32) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 t = reinterpret_cast<const float4*>(r)[i];
            v[4 * i + 0] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 32; i++) v[i] = r[min(i, n - 1)];
    }
}

// grid (ceil(cap/256), npairs); which = 0: query rows, 1: database rows.
// A thread owns a row and walks it in spans of 32 elements: the loads of a span
// are all in flight before the (strictly sequential) fmaf chain consumes them.
template <bool VEC>
__global__ __launch_bounds__(256) void k_row_norms(PairTable tbl, int which, int n_cap, int dim, int ld,
                                                   float* __restrict__ out, unsigned int* __restrict__ nmax)
{
    const mif_match_pair& pr = tbl[blockIdx.y];
    const float* __restrict__ m = which ? pr.d_db : pr.d_q;
    const int* d_n = which ? pr.d_nd : pr.d_nq;
    out += (size_t)blockIdx.y * n_cap;
    const int n = d_n ? min(*d_n, n_cap) : n_cap;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = m + (size_t)i * ld;
    float acc = 0.0f;
    for (int k0 = 0; k0 < dim; k0 += 32) {
        const int span = min(32, dim - k0);
        // a 16-byte load may run past `dim` as long as it stays inside the row
        const int avail = min(32, ld - k0);
        float v[32];
        load_span<VEC>(r + k0, avail, v);
#pragma unroll
        for (int k = 0; k < 32; k++)
            if (k < span) acc = __builtin_fmaf(v[k], v[k], acc);
    }
    out[i] = acc;
    // largest squared norm of the pair's rows (non-negative floats order like their bit patterns)
    if (nmax) atomicMax(&nmax[blockIdx.y], __builtin_bit_cast(unsigned int, acc));
}

struct Top2 {
    float s1, s2;
    int i1, i2;
};

__device__ __forceinline__ bool before(float sa, int ia, float sb, int ib)
{
    return sa < sb || (sa == sb && ia < ib);
}

// merge candidate (s, i) into a sorted pair
__device__ __forceinline__ void top2_insert(Top2& t, float s, int i)
{
    if (before(s, i, t.s1, t.i1)) {
        t.s2 = t.s1; t.i2 = t.i1;
        t.s1 = s; t.i1 = i;
    } else if (before(s, i, t.s2, t.i2)) {
        t.s2 = s; t.i2 = i;
    }
}

// Wave v of a block owns query rows [128 t + 32 v, +32) of query tile t; the
// block walks one split of the database in chunks of
// 64 rows (two 32-row MFMA tiles) staged in LDS and shared by the four waves.
//
// The queries are the B operand of the MFMA and live in registers for the whole
// kernel: lane l holds q[query l&31][2 kk + (l>>5)] for kk = 0 .. KSTEPS-1. The
// database chunk is the A operand, staged as s_b[buf][half][row][kk] =
// d[row][2 kk + half], so a lane fetches four k-steps with one ds_read_b128;
// rows are 16 B apart modulo the bank row (pitch KSTEPS + 4), which keeps those
// reads conflict free. The two tiles of a chunk give two independent
// accumulation chains that are issued alternately.
//
// Pipeline per chunk, one barrier: registers -> LDS buffer (c & 1) | barrier |
// issue the global loads of chunk c + 1 (not waited for until the next store) |
// MFMA over buffer (c & 1) | top-2 update. Staging loads are unconditional (row
// and column indices are clamped into the matrix, invalid values are replaced
// at the LDS store), so nothing between the loads and the MFMA loop waits.
//
// The grid is a fixed number of blocks that draw work items from a counter: an
// item is (pair, split of the database, tile of 128 queries), enumerated from
// the row counts in device memory, so pairs with few rows cost no idle blocks.
// Every block leaves the loop at the first item index past the total.
template <int KSTEPS, bool VEC>
__global__ __launch_bounds__(256) void k_match_partial(PairTable tbl, int npairs, int nq_cap, int nd_cap, int dim,
                                                       int ld, const float* __restrict__ qn_all,
                                                       const float* __restrict__ dn_all, int nsplit,
                                                       float* __restrict__ part_s_all, int* __restrict__ part_i_all,
                                                       unsigned int* __restrict__ work_counter)
{
    constexpr int K = 2 * KSTEPS;
    constexpr int KP = KSTEPS + 4;
    constexpr int K4 = K / 4;                  // float4 per staged row
    constexpr int NV = kDC * K4 / 256;         // float4 staged per thread and chunk
    constexpr int RSTEP = 256 / K4;            // rows between a thread's consecutive float4
    __shared__ __attribute__((aligned(16))) float s_b[2][2 * kDC * KP];
    __shared__ __attribute__((aligned(16))) float s_dn[2][kDC];
    __shared__ int s_first[kMaxPairs + 1];     // first item of every pair, then the total
    __shared__ int s_rows[kMaxPairs][2];       // query and database rows of every pair
    __shared__ int s_item;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;

    // Items of every pair (one thread per pair), then their exclusive prefix sums by the first wave:
    // s_first[z] = first item of pair z; pairs beyond npairs count nothing, s_first[kMaxPairs] = total.
    for (int z = tid; z < kMaxPairs; z += 256) {
        int cnt = 0;
        if (z < npairs) {
            const mif_match_pair& pz = tbl[z];
            const int nqz = pz.d_nq ? min(*pz.d_nq, nq_cap) : nq_cap;
            const int ndz = pz.d_nd ? min(*pz.d_nd, nd_cap) : nd_cap;
            s_rows[z][0] = nqz;
            s_rows[z][1] = ndz;
            const int chunks = (ndz + kDC - 1) / kDC;
            const int per = (chunks + nsplit - 1) / nsplit;
            const int live = per > 0 ? (chunks + per - 1) / per : 0;         // splits that hold rows
            cnt = ((nqz + kQT * kWaves - 1) / (kQT * kWaves)) * live;
        }
        s_first[z + 1] = cnt;
    }
    __syncthreads();
    if (tid < 64) {
        constexpr int PER = kMaxPairs / 64;
        int c[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { c[j] = s_first[tid * PER + j + 1]; sum += c[j]; }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (tid >= d) incl += o;
        }
        int run = incl - sum;
        if (tid == 0) s_first[0] = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { run += c[j]; s_first[tid * PER + j + 1] = run; }
        if (tid == 0) s_item = (int)atomicAdd(work_counter, 1u);
    }

    // staging coordinates of this thread: float4 column k4 of rows r0, r0 + RSTEP, ...
    const int k4 = tid % K4;
    const int r0 = tid / K4;
    const bool m0 = 4 * k4 + 0 < dim, m1 = 4 * k4 + 1 < dim, m2 = 4 * k4 + 2 < dim, m3 = 4 * k4 + 3 < dim;
    const int kc0 = min(4 * k4 + 0, dim - 1), kc1 = min(4 * k4 + 1, dim - 1);
    const int kc2 = min(4 * k4 + 2, dim - 1), kc3 = min(4 * k4 + 3, dim - 1);
    const int k4c = m0 ? 4 * k4 : 0;           // a 16-byte load must start inside the row

  for (;;) {
    __syncthreads();                           // s_item is set; the previous item is done with LDS
    const int item = __builtin_amdgcn_readfirstlane(s_item);
    __syncthreads();                           // every thread has read it
    if (item >= s_first[kMaxPairs]) break;     // uniform over the block
    // the item after this one is drawn now and published when this one is finished, so the
    // latency of the counter is covered by the work
    int next_item = 0;
    if (tid == 0) next_item = (int)atomicAdd(work_counter, 1u);
    // the pair that owns the item: the last z with s_first[z] <= item (empty pairs share their
    // first item with the next pair, so the last one among equals is the owner)
    int z = 0;
    {
        int hi = npairs - 1;
        while (z < hi) {
            const int mid = (z + hi + 1) >> 1;
            if (s_first[mid] <= item) z = mid; else hi = mid - 1;
        }
    }
    z = __builtin_amdgcn_readfirstlane(z);
    const mif_match_pair& pr = tbl[z];
    const float* __restrict__ Q = pr.d_q;
    const float* __restrict__ D = pr.d_db;
    const float* __restrict__ qn = qn_all + (size_t)z * nq_cap;
    const float* __restrict__ dn = dn_all + (size_t)z * nd_cap;
    float* __restrict__ part_s = part_s_all + (size_t)z * nsplit * nq_cap * 2;
    int* __restrict__ part_i = part_i_all + (size_t)z * nsplit * nq_cap * 2;
    const int nq = __builtin_amdgcn_readfirstlane(s_rows[z][0]);
    const int nd = __builtin_amdgcn_readfirstlane(s_rows[z][1]);
    const int qtiles = (nq + kQT * kWaves - 1) / (kQT * kWaves);
    const int split = (item - s_first[z]) / qtiles;             // consecutive items share database rows
    const int qtile = (item - s_first[z]) - split * qtiles;
    const int qbase = qtile * (kQT * kWaves) + wave * kQT;

    float qf[KSTEPS];                          // B fragments, filled below
    const int qi = qbase + col;
    float qnv = 0.0f;
    Top2 best;
    best.s1 = best.s2 = kInf;
    best.i1 = best.i2 = 0x7fffffff;

    const int chunks = (nd + kDC - 1) / kDC;
    const int per = (chunks + nsplit - 1) / nsplit;
    const int c_lo = split * per;
    const int c_hi = min(chunks, c_lo + per);


    float4 stage[NV];
    float stage_dn = 0.0f;
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int it = 0; it < NV; it++) {
            const int j = min(c * kDC + r0 + RSTEP * it, nd - 1);
            const float* src = D + (size_t)j * ld;
            if (VEC) {
                stage[it] = *reinterpret_cast<const float4*>(src + k4c);
            } else {
                stage[it] = make_float4(src[kc0], src[kc1], src[kc2], src[kc3]);
            }
        }
        stage_dn = dn[min(c * kDC + (tid & (kDC - 1)), nd - 1)];     // kept by the first kDC threads
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NV; it++) {
            const int r = r0 + RSTEP * it;
            const float4 v = stage[it];
            // k = 4 k4 + {0, 2} -> half 0, kk = 2 k4 + {0, 1};  k = 4 k4 + {1, 3} -> half 1
            *reinterpret_cast<float2*>(&s_b[buf][(0 * kDC + r) * KP + 2 * k4]) =
                make_float2(m0 ? v.x : 0.0f, m2 ? v.z : 0.0f);
            *reinterpret_cast<float2*>(&s_b[buf][(1 * kDC + r) * KP + 2 * k4]) =
                make_float2(m1 ? v.y : 0.0f, m3 ? v.w : 0.0f);
        }
        if (tid < kDC) s_dn[buf][tid] = stage_dn;
    };
    // scores of one result tile against the running top-2; database indices ascend with the
    // register index, and strict '<' keeps the lowest index on ties
    auto consider = [&](const f32x16& acc, int jbase, const float* dnp, bool full) {
        float sc[16];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 dn4 = *reinterpret_cast<const float4*>(&dnp[8 * g + 4 * half]);
            sc[4 * g + 0] = __builtin_fmaf(-2.0f, acc[4 * g + 0], qnv + dn4.x);
            sc[4 * g + 1] = __builtin_fmaf(-2.0f, acc[4 * g + 1], qnv + dn4.y);
            sc[4 * g + 2] = __builtin_fmaf(-2.0f, acc[4 * g + 2], qnv + dn4.z);
            sc[4 * g + 3] = __builtin_fmaf(-2.0f, acc[4 * g + 3], qnv + dn4.w);
        }
        if (!full) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                if (jbase + (r & 3) + 8 * (r >> 2) + 4 * half >= nd) sc[r] = kInf;
        }
        float mn = sc[0];
#pragma unroll
        for (int r = 1; r < 16; r++) mn = fminf(mn, sc[r]);
        if (mn < best.s2) {                     // rare once the top-2 has settled
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int j = jbase + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (sc[r] < best.s1) {
                    best.s2 = best.s1; best.i2 = best.i1;
                    best.s1 = sc[r]; best.i1 = j;
                } else if (sc[r] < best.s2) {
                    best.s2 = sc[r]; best.i2 = j;
                }
            }
        }
    };

    // one chunk of the pipeline; `buf` is a literal at both call sites so that every LDS
    // access of the body is a fixed offset from loop-invariant addresses
    auto chunk_step = [&](int c, int buf) {
        store_chunk(buf);
        // the buffer written here was last read two chunks ago, before the previous barrier
        __syncthreads();
        load_chunk(min(c + 1, c_hi - 1));      // unconditional: the last chunk is fetched twice
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        const float4* ap0 = reinterpret_cast<const float4*>(&s_b[buf][(half * kDC + col) * KP]);
        const float4* ap1 = reinterpret_cast<const float4*>(&s_b[buf][(half * kDC + 32 + col) * KP]);
#pragma unroll
        for (int q4 = 0; q4 < KSTEPS / 4; q4++) {
            const float4 a = ap0[q4];
            const float4 b = ap1[q4];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[4 * q4 + 0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, qf[4 * q4 + 0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[4 * q4 + 1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, qf[4 * q4 + 1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[4 * q4 + 2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, qf[4 * q4 + 2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[4 * q4 + 3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, qf[4 * q4 + 3], acc1, 0, 0, 0);
        }
        const bool full = (c + 1) * kDC <= nd;
        consider(acc0, c * kDC, &s_dn[buf][0], full);
        consider(acc1, c * kDC + 32, &s_dn[buf][32], full);
    };
    load_chunk(c_lo);                          // c_lo < c_hi for every enumerated item; in flight
                                               // while the queries are fetched
    // B fragments (queries): lane holds q[qbase + col][2 kk + half]
    {
        // every load is unconditional (row and column clamped into the matrix) and all of them
        // are in flight together; values outside the matrix are replaced afterwards
        const float* qr = Q + (size_t)min(qi, nq - 1) * ld;
        const bool row_ok = qi < nq;
        if (VEC) {
            // 16 bytes hold k = 4j .. 4j+3: this lane keeps the two of its half
            // (in two rounds, so that the raw rows do not crowd the register file)
            constexpr int HALF = K / 8;
#pragma unroll
            for (int h0 = 0; h0 < K / 4; h0 += HALF) {
                float rx[HALF], ry[HALF], rz[HALF], rw[HALF];
#pragma unroll
                for (int j = 0; j < HALF; j++) {
                    const float4 v = *reinterpret_cast<const float4*>(qr + min(4 * (h0 + j), (dim - 1) & ~3));
                    rx[j] = v.x; ry[j] = v.y; rz[j] = v.z; rw[j] = v.w;
                }
                asm volatile("" ::: "memory");  // keeps the loads above from sinking into the selects
#pragma unroll
                for (int j = 0; j < HALF; j++) {
                    const int k0 = 4 * (h0 + j);
                    const float e0 = half ? ry[j] : rx[j], e1 = half ? rw[j] : rz[j];
                    qf[2 * (h0 + j) + 0] = (row_ok && k0 + 0 + half < dim) ? e0 : 0.0f;
                    qf[2 * (h0 + j) + 1] = (row_ok && k0 + 2 + half < dim) ? e1 : 0.0f;
                }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KSTEPS; kk++) qf[kk] = qr[min(2 * kk + half, dim - 1)];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < KSTEPS; kk++) qf[kk] = (row_ok && 2 * kk + half < dim) ? qf[kk] : 0.0f;
        }
    }
    // A 32x32 result tile has the database row on the register index and the
    // query on the lane: C[(reg&3) + 8 (reg>>2) + 4 half][col]. A lane therefore
    // keeps ONE running top-2 for its query; lanes l and l + 32 hold the same
    // query and are merged at the end.
    qnv = (qi < nq) ? qn[qi] : 0.0f;

    for (int c = c_lo; c < c_hi; c += 2) {
        chunk_step(c, 0);
        if (c + 1 < c_hi) chunk_step(c + 1, 1);
    }
    // merge the two halves of the wave (same query, interleaved database rows)
    {
        const float os1 = __shfl_xor(best.s1, 32), os2 = __shfl_xor(best.s2, 32);
        const int oi1 = __shfl_xor(best.i1, 32), oi2 = __shfl_xor(best.i2, 32);
        top2_insert(best, os1, oi1);
        top2_insert(best, os2, oi2);
        if (half == 0 && qi < nq) {
            const size_t o = ((size_t)split * nq_cap + qi) * 2;
            part_s[o] = best.s1; part_s[o + 1] = best.s2;
            part_i[o] = best.i1; part_i[o + 1] = best.i2;
        }
    }
    if (tid == 0) s_item = next_item;
  }
}

// ---------------------------------------------------------------- screening
//
// The exact search above spends its time in fp32 matrix instructions. For batches of small problems
// (frame against frame) the same answer comes much cheaper: bf16 matrix instructions run 16 times
// as fast, and a bf16 score s~ differs from the fp32 ranking score s by at most a margin m that
// follows from the norms (kScreenRel, kScreenAbs). If t is the second smallest s~ of a query, every
// row among the query's true two nearest has s~ <= t + 2 m: the two rows with the smallest s~ have
// s <= t + m, so the second smallest s is <= t + m, and a row with s <= t + m has s~ <= t + 2 m.
//   pass 0: t per query and database split (branch-free running two smallest)
//   pass 1: threshold = smallest t of the query's splits + 2 m; the rows with s~ <= threshold are
//     appended to the candidate list of (query, split)
//   k_match_rank: fp32 ranking scores of the candidates, computed exactly as the exact search does
//     (fmaf chain in ascending k, fmaf(-2, dot, |q|^2 + |d|^2)), two smallest (score, index); a
//     list that overflowed stands for all rows of its split, and a query whose lists hold fewer
//     rows than the answer has is ranked against all rows.
// The result is therefore the exact search's, bit for bit; the bf16 values only decide which rows
// are looked at. Same block structure as k_match_partial: 4 waves x 32 queries, database chunks of
// 64 rows staged in LDS (as bf16, converted while staging), persistent blocks drawing
// (pair, database split, 128-query tile) items.
template <int K, bool VEC, int PASS>
__global__ __launch_bounds__(256) void k_match_screen(PairTable tbl, int npairs, int nq_cap, int nd_cap, int dim,
                                                      int ld, const float* __restrict__ qn_all,
                                                      const float* __restrict__ dn_all,
                                                      const unsigned int* __restrict__ dnmax_all, int nsplit,
                                                      float* __restrict__ t2_all, int* __restrict__ cnt_all,
                                                      int2* __restrict__ cand_all,
                                                      unsigned int* __restrict__ work_counter)
{
    constexpr int KS = K / 16;                 // matrix instructions per tile
    constexpr int KPB = K + 8;                 // row pitch in bf16 elements (16 bytes of padding)
    constexpr int K4 = K / 4;                  // float4 per staged row
    constexpr int NV = kDC * K4 / 256;         // float4 staged per thread and chunk
    constexpr int RSTEP = 256 / K4;            // rows between a thread's consecutive float4
    __shared__ __attribute__((aligned(16))) __bf16 s_b[2][kDC * KPB];
    __shared__ __attribute__((aligned(16))) float s_dn[2][kDC];
    __shared__ int s_first[kMaxPairs + 1];
    __shared__ int s_rows[kMaxPairs][2];
    __shared__ int s_item;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;

    for (int z = tid; z < kMaxPairs; z += 256) {
        int cnt = 0;
        if (z < npairs) {
            const mif_match_pair& pz = tbl[z];
            const int nqz = pz.d_nq ? min(*pz.d_nq, nq_cap) : nq_cap;
            const int ndz = pz.d_nd ? min(*pz.d_nd, nd_cap) : nd_cap;
            s_rows[z][0] = nqz;
            s_rows[z][1] = ndz;
            const int chunks = (ndz + kDC - 1) / kDC;
            const int per = (chunks + nsplit - 1) / nsplit;
            const int live = per > 0 ? (chunks + per - 1) / per : 0;         // splits that hold rows
            cnt = ((nqz + kQT * kWaves - 1) / (kQT * kWaves)) * live;
        }
        s_first[z + 1] = cnt;
    }
    __syncthreads();
    if (tid < 64) {
        constexpr int PER = kMaxPairs / 64;
        int c[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { c[j] = s_first[tid * PER + j + 1]; sum += c[j]; }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (tid >= d) incl += o;
        }
        int run = incl - sum;
        if (tid == 0) s_first[0] = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { run += c[j]; s_first[tid * PER + j + 1] = run; }
        if (tid == 0) s_item = (int)atomicAdd(work_counter, 1u);
    }

    const int k4 = tid % K4;
    const int r0 = tid / K4;
    const bool m0 = 4 * k4 + 0 < dim, m1 = 4 * k4 + 1 < dim, m2 = 4 * k4 + 2 < dim, m3 = 4 * k4 + 3 < dim;
    const int kc0 = min(4 * k4 + 0, dim - 1), kc1 = min(4 * k4 + 1, dim - 1);
    const int kc2 = min(4 * k4 + 2, dim - 1), kc3 = min(4 * k4 + 3, dim - 1);
    const int k4c = m0 ? 4 * k4 : 0;

  for (;;) {
    __syncthreads();
    const int item = __builtin_amdgcn_readfirstlane(s_item);
    __syncthreads();
    if (item >= s_first[kMaxPairs]) break;     // uniform over the block
    int next_item = 0;
    if (tid == 0) next_item = (int)atomicAdd(work_counter, 1u);
    int z = 0;
    {
        int hi = npairs - 1;
        while (z < hi) {
            const int mid = (z + hi + 1) >> 1;
            if (s_first[mid] <= item) z = mid; else hi = mid - 1;
        }
    }
    z = __builtin_amdgcn_readfirstlane(z);
    const mif_match_pair& pr = tbl[z];
    const float* __restrict__ Q = pr.d_q;
    const float* __restrict__ D = pr.d_db;
    const float* __restrict__ qn = qn_all + (size_t)z * nq_cap;
    const float* __restrict__ dn = dn_all + (size_t)z * nd_cap;
    const int nq = __builtin_amdgcn_readfirstlane(s_rows[z][0]);
    const int nd = __builtin_amdgcn_readfirstlane(s_rows[z][1]);
    const int qtiles = (nq + kQT * kWaves - 1) / (kQT * kWaves);
    const int split = (item - s_first[z]) / qtiles;             // consecutive items share database rows
    const int qbase = ((item - s_first[z]) - split * qtiles) * (kQT * kWaves) + wave * kQT;
    const int qi = qbase + col;
    const int chunks = (nd + kDC - 1) / kDC;
    const int per = (chunks + nsplit - 1) / nsplit;
    const int c_lo = split * per;
    const int c_hi = min(chunks, c_lo + per);
    // per (pair, split, query): second smallest score of the split (pass 0), candidate count and list (pass 1)
    float* __restrict__ t2s = t2_all + (size_t)z * nsplit * nq_cap;
    int* __restrict__ cnt = cnt_all + ((size_t)z * nsplit + split) * nq_cap;
    int2* __restrict__ cand = cand_all + ((size_t)z * nsplit + split) * nq_cap * kCandCap;

    float4 stage[NV];
    float stage_dn = 0.0f;
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int it = 0; it < NV; it++) {
            const int j = min(c * kDC + r0 + RSTEP * it, nd - 1);
            const float* src = D + (size_t)j * ld;
            if (VEC) stage[it] = *reinterpret_cast<const float4*>(src + k4c);
            else stage[it] = make_float4(src[kc0], src[kc1], src[kc2], src[kc3]);
        }
        stage_dn = dn[min(c * kDC + (tid & (kDC - 1)), nd - 1)];
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NV; it++) {
            const int r = r0 + RSTEP * it;
            const float4 v = stage[it];
            bf16x4 b;
            b[0] = (__bf16)(m0 ? v.x : 0.0f); b[1] = (__bf16)(m1 ? v.y : 0.0f);
            b[2] = (__bf16)(m2 ? v.z : 0.0f); b[3] = (__bf16)(m3 ? v.w : 0.0f);
            *reinterpret_cast<bf16x4*>(&s_b[buf][r * KPB + 4 * k4]) = b;
        }
        if (tid < kDC) s_dn[buf][tid] = stage_dn;
    };
    load_chunk(c_lo);

    // queries (B operand): lane holds q[qbase + col][16 j + 8 half .. + 7] for every instruction j
    bf16x8 qf[KS];
    {
        const float* qr = Q + (size_t)min(qi, nq - 1) * ld;
        const bool row_ok = qi < nq;
#pragma unroll
        for (int j = 0; j < KS; j++) {
            const int k0 = 16 * j + 8 * half;
            float e[8];
            if (VEC) {
                const float4 u = *reinterpret_cast<const float4*>(qr + min(k0, (dim - 1) & ~3));
                const float4 w = *reinterpret_cast<const float4*>(qr + min(k0 + 4, (dim - 1) & ~3));
                e[0] = u.x; e[1] = u.y; e[2] = u.z; e[3] = u.w; e[4] = w.x; e[5] = w.y; e[6] = w.z; e[7] = w.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) e[i] = qr[min(k0 + i, dim - 1)];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) qf[j][i] = (__bf16)((row_ok && k0 + i < dim) ? e[i] : 0.0f);
        }
    }
    const float qnv = (qi < nq) ? qn[qi] : 0.0f;
    float t1 = kInf, t2 = kInf;                 // pass 0: the two smallest scores seen
    // pass 1: the query's threshold. The second smallest score over all rows is at most the second
    // smallest of any one split, so the smallest of the splits' values (+ 2 m) keeps every winner.
    float limit = -kInf;
    if (PASS == 1 && qi < nq) {
        float t = kInf;
        for (int sp = 0; sp * per < chunks; sp++) t = fminf(t, t2s[(size_t)sp * nq_cap + qi]);
        const float dmax = __builtin_bit_cast(float, dnmax_all[z]);
        const float m = kScreenRel * __builtin_sqrtf(qnv * dmax) + kScreenAbs * (qnv + dmax);
        limit = t + 2.0f * m;                   // +inf while no split holds two rows: every row is kept
    }
    int filled = 0;                             // pass 1: rows appended to the query's list so far

    auto consider = [&](const f32x16& acc, int jbase, const float* dnp, bool full) {
        float sc[16];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 dn4 = *reinterpret_cast<const float4*>(&dnp[8 * g + 4 * half]);
            sc[4 * g + 0] = __builtin_fmaf(-2.0f, acc[4 * g + 0], qnv + dn4.x);
            sc[4 * g + 1] = __builtin_fmaf(-2.0f, acc[4 * g + 1], qnv + dn4.y);
            sc[4 * g + 2] = __builtin_fmaf(-2.0f, acc[4 * g + 2], qnv + dn4.z);
            sc[4 * g + 3] = __builtin_fmaf(-2.0f, acc[4 * g + 3], qnv + dn4.w);
        }
        if (!full) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                if (jbase + (r & 3) + 8 * (r >> 2) + 4 * half >= nd) sc[r] = kInf;
        }
        if (PASS == 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                t2 = __builtin_amdgcn_fmed3f(t1, t2, sc[r]);     // t1 <= t2: the middle one is the new second
                t1 = fminf(t1, sc[r]);
            }
        } else {
            float mn = sc[0];
#pragma unroll
            for (int r = 1; r < 16; r++) mn = fminf(mn, sc[r]);
            // A query's rows are seen by the two lanes l, l + 32 of this wave only, so its list is
            // filled without atomics: the branch is uniform over the wave, the lower lane's rows of
            // a tile go first, and both lanes keep the same running count.
            if (__builtin_amdgcn_ballot_w64(mn <= limit) != 0ull) {
                unsigned int hits = 0u;
#pragma unroll
                for (int r = 0; r < 16; r++) hits |= (sc[r] <= limit ? 1u : 0u) << r;
                const int mine = __popc(hits);
                const int other = __shfl_xor(mine, 32);
                int at = filled + (half ? other : 0);
                filled += mine + other;
                while (hits != 0u) {
                    const int r = __ffs((int)hits) - 1;
                    hits &= hits - 1u;
                    if (at < kCandCap) cand[(size_t)qi * kCandCap + at] = make_int2(jbase + (r & 3) + 8 * (r >> 2) + 4 * half, 0);
                    at++;
                }
            }
        }
    };
    auto chunk_step = [&](int c, int buf) {
        store_chunk(buf);
        __syncthreads();
        load_chunk(min(c + 1, c_hi - 1));
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        const __bf16* ap0 = &s_b[buf][col * KPB + 8 * half];
        const __bf16* ap1 = &s_b[buf][(32 + col) * KPB + 8 * half];
#pragma unroll
        for (int j = 0; j < KS; j++) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap0 + 16 * j);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(ap1 + 16 * j);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, qf[j], acc1, 0, 0, 0);
        }
        const bool full = (c + 1) * kDC <= nd;
        consider(acc0, c * kDC, &s_dn[buf][0], full);
        consider(acc1, c * kDC + 32, &s_dn[buf][32], full);
    };
    for (int c = c_lo; c < c_hi; c += 2) {
        chunk_step(c, 0);
        if (c + 1 < c_hi) chunk_step(c + 1, 1);
    }
    if (PASS == 0) {
        // lanes l and l + 32 hold the same query (interleaved database rows)
        const float o1 = __shfl_xor(t1, 32), o2 = __shfl_xor(t2, 32);
        t2 = __builtin_amdgcn_fmed3f(t1, t2, o1); t1 = fminf(t1, o1);
        t2 = __builtin_amdgcn_fmed3f(t1, t2, o2); t1 = fminf(t1, o2);
        if (half == 0 && qi < nq) t2s[(size_t)split * nq_cap + qi] = t2;
    } else if (half == 0 && qi < nq) {
        cnt[qi] = filled;
    }
    if (tid == 0) s_item = next_item;
  }
}

// The screening passes again, arranged for the matrix cores' appetite (the form used by default):
//  * a wave owns 64 queries (two B operands in registers), a block 256: every database row staged in
//    LDS and every A operand read from it now feeds twice the matrix instructions, so the passes ask
//    half as much of the L2 and of the LDS per score;
//  * the database norm rides in the contraction: the rows get 16 more k-elements, three of which hold
//    -|d|^2 / 2 split into bf16 parts (exact to 2^-27 of it), and the queries hold 1 there, so a matrix
//    result is a = q.d - |d|^2 / 2 and the screening score is |q|^2 - 2 a. The epilogue thus only tracks
//    the two LARGEST a per query (v_max + v_med3 per value) instead of building every score first
//    (add + fma) and reading the norms from LDS. Rows beyond the database end carry -3e38 there and
//    never win.
// Same items otherwise (pair, database split, query tile), same per-(pair, split, query) outputs as
// k_match_screen: the second smallest score (pass 0), the candidate list and its count (pass 1).
constexpr int kQW2 = 64;                       // queries per wave
constexpr float kPadNorm = -3.0e38f;           // norm part of the rows staged beyond the database end

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef const float __attribute__((address_space(1)))* GlobalF32;
typedef const f32x4v __attribute__((address_space(1)))* GlobalF32x4;

// The larger of two finite numbers in one instruction: fmaxf would first canonicalise both operands
// (one more instruction each), which values that only ever meet comparisons do not need. (Not inline
// assembly: the compiler does not track the wait states between a matrix instruction and an assembly
// statement that reads its result.)
__device__ __forceinline__ float max_raw(float a, float b)
{
    return __builtin_amdgcn_fmed3f(a, b, 3.4028234663852886e38f);
}

// WV waves per block: 4 (256 queries per work item; two blocks per compute unit) for batches of small
// searches, 8 (512 queries, one block per compute unit) for large ones, where every staged database row
// then serves twice the queries and the blocks ask half as much of the L2.
// Between the two passes of k_match_screen2 / k_match_screen4: per (pair, query) the limit of pass 1
// from the bounds pass 0 left per split, written over the bound of split 0. A work item of pass 1
// (one per split and query tile) then reads one value per query instead of walking all splits: with 64
// splits that walk, a chain of dependent loads at the head of every item, took a quarter of the pass.
// A row is kept when a >= (|q|^2 - (t + 2 m)) / 2, t a bound on the second smallest score (below):
// +inf while the database holds fewer than two rows, so that every row is kept.
// grid (ceil(nq_cap / 256), pairs).
__global__ __launch_bounds__(256) void k_match_limits(PairTable tbl, int nq_cap, int nd_cap,
                                                      const float* __restrict__ qn_all,
                                                      const unsigned int* __restrict__ dnmax_all, int nsplit,
                                                      float* __restrict__ t2_all, size_t s1_off)
{
    const int z = blockIdx.y;
    const mif_match_pair& pz = tbl[z];
    const int nq = pz.d_nq ? min(*pz.d_nq, nq_cap) : nq_cap;
    const int nd = pz.d_nd ? min(*pz.d_nd, nd_cap) : nd_cap;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const int chunks = (nd + kDC - 1) / kDC;
    const int per = (chunks + nsplit - 1) / nsplit;
    float* __restrict__ t2s = t2_all + (size_t)z * nsplit * nq_cap;
    // the second smallest score over all rows is at most the bound any split found for its own second
    // smallest, and at most the second smallest of the splits' smallest scores (two different rows)
    float t = kInf, b1 = kInf, b2 = kInf;
    for (int sp = 0; sp * per < chunks; sp++) {
        t = fminf(t, t2s[(size_t)sp * nq_cap + i]);
        const float v = t2s[s1_off + (size_t)sp * nq_cap + i];
        b2 = fminf(b2, fmaxf(b1, v));
        b1 = fminf(b1, v);
    }
    t = fminf(t, b2);
    const float qnv = qn_all[(size_t)z * nq_cap + i];
    const float dmax = __builtin_bit_cast(float, dnmax_all[z]);
    const float m = kScreenRel * __builtin_sqrtf(qnv * dmax) + 3.0f * kScreenAbs * (qnv + dmax);
    const float limit = t + 2.0f * m;
    t2s[i] = 0.5f * (qnv - limit);
}

template <int K, bool VEC, int PASS, int WV>
__global__ __launch_bounds__(64 * WV, 8 / WV) void k_match_screen2(PairTable tbl, int npairs, int nq_cap, int nd_cap, int dim,
                                                       int ld, const float* __restrict__ qn_all,
                                                       const float* __restrict__ dn_all,
                                                       const unsigned int* __restrict__ dnmax_all, int nsplit,
                                                       float* __restrict__ t2_all, size_t s1_off,
                                                       int* __restrict__ cnt_all, int2* __restrict__ cand_all,
                                                       unsigned int* __restrict__ work_counter, int sample)
{
    constexpr int KP = K + 16;                 // contraction length with the norm's k-step
    constexpr int KS = KP / 16;                // matrix instructions per tile
    constexpr int KPB = KP + 8;                // row pitch in bf16 elements (16 bytes of padding)
    constexpr int K4 = K / 4;                  // float4 per staged row
    constexpr int NT = 64 * WV;                // threads per block
    constexpr int kQB2 = kQW2 * WV;            // queries per block and work item
    constexpr int NV = kDC * K4 / NT;          // float4 staged per thread and chunk
    constexpr int RSTEP = NT / K4;             // rows between a thread's consecutive float4
    static_assert(NV >= 1 && NV * NT == kDC * K4, "staging: whole float4 per thread");
    __shared__ __attribute__((aligned(16))) __bf16 s_b[2][kDC * KPB];
    __shared__ int s_first[kMaxPairs + 1];
    __shared__ int s_rows[kMaxPairs][2];
    __shared__ int s_item;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;

    for (int z = tid; z < kMaxPairs; z += NT) {
        int cnt = 0;
        if (z < npairs) {
            const mif_match_pair& pz = tbl[z];
            const int nqz = pz.d_nq ? min(*pz.d_nq, nq_cap) : nq_cap;
            const int ndz = pz.d_nd ? min(*pz.d_nd, nd_cap) : nd_cap;
            s_rows[z][0] = nqz;
            s_rows[z][1] = ndz;
            const int chunks = (ndz + kDC - 1) / kDC;
            const int per = (chunks + nsplit - 1) / nsplit;
            const int live = per > 0 ? (chunks + per - 1) / per : 0;         // splits that hold rows
            cnt = ((nqz + kQB2 - 1) / kQB2) * live;
        }
        s_first[z + 1] = cnt;
    }
    __syncthreads();
    if (tid < 64) {
        constexpr int PER = kMaxPairs / 64;
        int c[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { c[j] = s_first[tid * PER + j + 1]; sum += c[j]; }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (tid >= d) incl += o;
        }
        int run = incl - sum;
        if (tid == 0) s_first[0] = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { run += c[j]; s_first[tid * PER + j + 1] = run; }
        if (tid == 0) s_item = (int)atomicAdd(work_counter, 1u);
    }

    const int k4 = tid % K4;
    const int r0 = tid / K4;
    const bool m0 = 4 * k4 + 0 < dim, m1 = 4 * k4 + 1 < dim, m2 = 4 * k4 + 2 < dim, m3 = 4 * k4 + 3 < dim;
    const int kc0 = min(4 * k4 + 0, dim - 1), kc1 = min(4 * k4 + 1, dim - 1);
    const int kc2 = min(4 * k4 + 2, dim - 1), kc3 = min(4 * k4 + 3, dim - 1);
    const int k4c = m0 ? 4 * k4 : 0;

  for (;;) {
    __syncthreads();
    const int item = __builtin_amdgcn_readfirstlane(s_item);
    __syncthreads();
    if (item >= s_first[kMaxPairs]) break;     // uniform over the block
    int next_item = 0;
    if (tid == 0) next_item = (int)atomicAdd(work_counter, 1u);
    int z = 0;
    {
        int hi = npairs - 1;
        while (z < hi) {
            const int mid = (z + hi + 1) >> 1;
            if (s_first[mid] <= item) z = mid; else hi = mid - 1;
        }
    }
    z = __builtin_amdgcn_readfirstlane(z);
    const mif_match_pair& pr = tbl[z];
    const float* __restrict__ Q = pr.d_q;
    const float* __restrict__ D = pr.d_db;
    const float* __restrict__ qn = qn_all + (size_t)z * nq_cap;
    const float* __restrict__ dn = dn_all + (size_t)z * nd_cap;
    const int nq = __builtin_amdgcn_readfirstlane(s_rows[z][0]);
    const int nd = __builtin_amdgcn_readfirstlane(s_rows[z][1]);
    const int qtiles = (nq + kQB2 - 1) / kQB2;
    const int split = (item - s_first[z]) / qtiles;             // consecutive items share database rows
    const int qbase = ((item - s_first[z]) - split * qtiles) * kQB2 + wave * kQW2;
    const int chunks = (nd + kDC - 1) / kDC;
    const int per = (chunks + nsplit - 1) / nsplit;
    const int c_lo = split * per;
    const int c_hi = min(chunks, c_lo + per);
    float* __restrict__ t2s = t2_all + (size_t)z * nsplit * nq_cap;
    int* __restrict__ cnt = cnt_all + ((size_t)z * nsplit + split) * nq_cap;
    int2* __restrict__ cand = cand_all + ((size_t)z * nsplit + split) * nq_cap * kCandCap;

    f32x4v stage[NV];
    float stage_dn = 0.0f;
    bool stage_pad = false;
    // (loads through global-address-space pointers: as generic pointers they would be flat loads, which
    // also count as LDS operations and so make every wait for an LDS read wait for them too)
    const GlobalF32 Dg = (GlobalF32)D;
    const GlobalF32 dng = (GlobalF32)dn;
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int it = 0; it < NV; it++) {
            const int j = min(c * kDC + r0 + RSTEP * it, nd - 1);
            const GlobalF32 src = Dg + (size_t)j * ld;
            if (VEC) stage[it] = *reinterpret_cast<GlobalF32x4>(src + k4c);
            else stage[it] = (f32x4v){src[kc0], src[kc1], src[kc2], src[kc3]};
        }
        // (only loaded here; anything computed from it now would wait for all the loads above)
        const int jr = c * kDC + (tid & (kDC - 1));
        stage_dn = dng[min(jr, nd - 1)];
        stage_pad = jr >= nd;
    };
    // elements at and beyond dim are cleared after the conversion (two packed ANDs per four elements)
    const uint32_t and01 = (m0 ? 0x0000ffffu : 0u) | (m1 ? 0xffff0000u : 0u);
    const uint32_t and23 = (m2 ? 0x0000ffffu : 0u) | (m3 ? 0xffff0000u : 0u);
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NV; it++) {
            const int r = r0 + RSTEP * it;
            uint2 u = __builtin_bit_cast(uint2, __builtin_convertvector(stage[it], bf16x4));
            u.x &= and01; u.y &= and23;
            *reinterpret_cast<uint2*>(&s_b[buf][r * KPB + 4 * k4]) = u;
        }
        if (tid < kDC) {
            // the norm's k-step: -|d|^2 / 2 as three bf16 parts, then zeros
            const float hn = stage_pad ? kPadNorm : -0.5f * stage_dn;
            const __bf16 h1 = (__bf16)hn;
            const float r1 = hn - (float)h1;
            const __bf16 h2 = (__bf16)r1;
            const __bf16 h3 = (__bf16)(r1 - (float)h2);
            bf16x8 a, zz;
#pragma unroll
            for (int i = 0; i < 8; i++) { a[i] = (__bf16)0.0f; zz[i] = (__bf16)0.0f; }
            a[0] = h1; a[1] = h2; a[2] = h3;
            *reinterpret_cast<bf16x8*>(&s_b[buf][tid * KPB + K]) = a;
            *reinterpret_cast<bf16x8*>(&s_b[buf][tid * KPB + K + 8]) = zz;
        }
    };
    // pass 0 may look at every `sample`-th chunk only (see k_match_screen4)
    const int cs = PASS == 0 ? sample : 1;
    const int c_last = c_lo + (c_hi - 1 - c_lo) / cs * cs;
    load_chunk(c_lo);

    // queries (B operands): lane holds q[qbase + 32 s + col][16 j + 8 half .. + 7] for every instruction j
    bf16x8 qf[2][KS];
    float qnv[2];
    int qi[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
        qi[s2] = qbase + 32 * s2 + col;
        const float* qr = Q + (size_t)min(qi[s2], nq - 1) * ld;
        const bool row_ok = qi[s2] < nq;
#pragma unroll
        for (int j = 0; j < KS - 1; j++) {
            const int k0 = 16 * j + 8 * half;
            float e[8];
            if (VEC) {
                const float4 u = *reinterpret_cast<const float4*>(qr + min(k0, (dim - 1) & ~3));
                const float4 w = *reinterpret_cast<const float4*>(qr + min(k0 + 4, (dim - 1) & ~3));
                e[0] = u.x; e[1] = u.y; e[2] = u.z; e[3] = u.w; e[4] = w.x; e[5] = w.y; e[6] = w.z; e[7] = w.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) e[i] = qr[min(k0 + i, dim - 1)];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) qf[s2][j][i] = (__bf16)((row_ok && k0 + i < dim) ? e[i] : 0.0f);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) qf[s2][KS - 1][i] = (__bf16)((row_ok && half == 0 && i < 3) ? 1.0f : 0.0f);
        qnv[s2] = row_ok ? qn[qi[s2]] : 0.0f;
    }
    float t1[2] = {-kInf, -kInf}, t2[2] = {-kInf, -kInf};     // pass 0: the two largest a seen
    // pass 1: a row is kept when its score |q|^2 - 2 a is at most the query's threshold, i.e. when
    // a >= (|q|^2 - threshold) / 2. The second smallest score over all rows is at most the second
    // smallest of any one split, so the smallest of the splits' values (+ 2 m) keeps every winner.
    float alim[2] = {kInf, kInf};
    int filled[2] = {0, 0};
    if (PASS == 1) {            // (k_match_limits left the query's limit in the slot of split 0)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
            if (qi[s2] < nq) alim[s2] = t2s[qi[s2]];
    }

    auto consider = [&](const f32x16& acc, int jbase, int s2, bool full) {
        if (PASS == 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                t2[s2] = __builtin_amdgcn_fmed3f(t1[s2], t2[s2], acc[r]);     // t1 >= t2: the middle one is the new second
                t1[s2] = max_raw(t1[s2], acc[r]);
            }
        } else {
            float mx = acc[0];
#pragma unroll
            for (int r = 1; r < 16; r++) mx = max_raw(mx, acc[r]);
            // A query's rows are seen by the two lanes l, l + 32 of this wave only, so its list is
            // filled without atomics: the branch is uniform over the wave, the lower lane's rows of
            // a tile go first, and both lanes keep the same running count.
            if (__builtin_amdgcn_ballot_w64(mx >= alim[s2]) != 0ull) {
                unsigned int hits = 0u;
#pragma unroll
                for (int r = 0; r < 16; r++) hits |= (acc[r] >= alim[s2] ? 1u : 0u) << r;
                // What goes into the list with a row: the largest product of this lane's 16 rows, which is the
                // product of the best row kept here and an upper bound for the others (k_match_rank tightens
                // the threshold with these before it reads any row).
                float top = mx;
                if (!full) {                    // rows staged beyond the database end (they pass an infinite threshold)
                    top = -kInf;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        if (jbase + (r & 3) + 8 * (r >> 2) + 4 * half >= nd) hits &= ~(1u << r);
                        else top = fmaxf(top, acc[r]);
                    }
                }
                const int topbits = __float_as_int(top);
                const int mine = __popc(hits);
                const int other = __shfl_xor(mine, 32);
                int at = filled[s2] + (half ? other : 0);
                filled[s2] += mine + other;
                while (hits != 0u) {
                    const int r = __ffs((int)hits) - 1;
                    hits &= hits - 1u;
                    if (at < kCandCap)
                        cand[(size_t)qi[s2] * kCandCap + at] = make_int2(jbase + (r & 3) + 8 * (r >> 2) + 4 * half, topbits);
                    at++;
                }
            }
        }
    };
    auto chunk_step = [&](int c, int buf) {
        store_chunk(buf);
        __syncthreads();
        load_chunk(min(c + cs, c_last));
        f32x16 acc00, acc01, acc10, acc11;      // [row tile][query set]
#pragma unroll
        for (int r = 0; r < 16; r++) { acc00[r] = 0.0f; acc01[r] = 0.0f; acc10[r] = 0.0f; acc11[r] = 0.0f; }
        const __bf16* ap0 = &s_b[buf][col * KPB + 8 * half];
        const __bf16* ap1 = &s_b[buf][(32 + col) * KPB + 8 * half];
        // the A operands of k-step j + 1 are read from LDS while the matrix instructions of step j run
        bf16x8 a = *reinterpret_cast<const bf16x8*>(ap0);
        bf16x8 b = *reinterpret_cast<const bf16x8*>(ap1);
#pragma unroll
        for (int j = 0; j < KS; j++) {
            bf16x8 an = a, bn = b;
            if (j + 1 < KS) {
                an = *reinterpret_cast<const bf16x8*>(ap0 + 16 * (j + 1));
                bn = *reinterpret_cast<const bf16x8*>(ap1 + 16 * (j + 1));
            }
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[0][j], acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[1][j], acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, qf[0][j], acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, qf[1][j], acc11, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // the two LDS reads of the next step ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // ... then this step's four matrix instructions
            a = an; b = bn;
        }
        const bool full = (c + 1) * kDC <= nd;
        consider(acc00, c * kDC, 0, full);
        consider(acc01, c * kDC, 1, full);
        consider(acc10, c * kDC + 32, 0, full);
        consider(acc11, c * kDC + 32, 1, full);
    };
    for (int c = c_lo; c < c_hi; c += 2 * cs) {
        chunk_step(c, 0);
        if (c + cs < c_hi) chunk_step(c + cs, 1);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
        if (PASS == 0) {
            // lanes l and l + 32 hold the same query (interleaved database rows)
            const float o1 = __shfl_xor(t1[s2], 32), o2 = __shfl_xor(t2[s2], 32);
            float a1 = t1[s2], a2 = t2[s2];
            a2 = __builtin_amdgcn_fmed3f(a1, a2, o1); a1 = fmaxf(a1, o1);
            a2 = __builtin_amdgcn_fmed3f(a1, a2, o2); a1 = fmaxf(a1, o2);
            // as scores; +inf when the split holds fewer rows (-inf or a padded row's -3e38)
            if (half == 0 && qi[s2] < nq) {
                t2s[(size_t)split * nq_cap + qi[s2]] = __builtin_fmaf(-2.0f, a2, qnv[s2]);
                t2s[s1_off + (size_t)split * nq_cap + qi[s2]] = __builtin_fmaf(-2.0f, a1, qnv[s2]);
            }
        } else if (half == 0 && qi[s2] < nq) {
            cnt[qi[s2]] = filled[s2];
        }
    }
    if (tid == 0) s_item = next_item;
  }
}

// ---- screening on a bf16 copy of the database, staged by LDS-DMA, the waves of a SIMD working in turns
//
// In k_match_screen2 the two waves that share a SIMD (one of each of a compute unit's two blocks) fall into
// step: both issue their matrix instructions at the same time, at half speed each, and then both run
// their vector work while the matrix pipe idles (42 % of the wave cycles waiting for an issue slot, the
// pipe busy 47 %). Here a block has 8 waves, two per SIMD (waves w and w + 4 share one), and its halves
// alternate between a matrix phase (the matrix instructions of a chunk) and a vector phase (the epilogue
// of their previous chunk), half a chunk apart, with a barrier between the half periods:
//     half period 2c     : waves 0-3 matrix(c)   | waves 4-7 epilogue(c - 1)
//     half period 2c + 1 : waves 0-3 epilogue(c) | waves 4-7 matrix(c)
// The vector phase has to be short for that, and the registers few (two waves per SIMD leave 256 each,
// and 64 accumulators + 72 of B operands are given): so the staging does not go through registers. The
// database rows are converted ONCE per search into a bf16 image that is what a chunk looks like in LDS
// (k_match_pack_rows: pitch K + 24 elements, the norm's k-step included, rows padded to whole chunks), and
// a chunk is then 19 (11 for K = 64) wave-wide 1 KB copies straight into LDS (global_load_lds_dwordx4: the
// LDS address is a wave-uniform base + 16 bytes per lane, the image is contiguous on both sides), issued
// at the start of half period 2c for chunk c + 1 and waited for before the barrier that ends 2c + 1.
// All LDS of the kernel is one array: a second __shared__ object beside a DMA target can make the
// compiler wait for the DMA before every LDS read. The barriers of the loop are raw s_barrier: a
// __syncthreads() would drain the copy in flight.

// The bf16 image of the database rows of every pair: [pair][rows_pad][K + 24], rows_pad a multiple of the
// chunk size. Element k < K: bf16(d[k]) (0 at and beyond dim); K .. K + 2: -|d|^2 / 2 in three bf16 parts
// (kPadNorm for rows at and beyond the pair's row count, which therefore never win); the rest 0.
// One thread per 16-byte piece of a row. grid (ceil(rows_pad * pieces / 256), pairs).
template <int K, bool VEC>
__global__ __launch_bounds__(256) void k_match_pack_rows(PairTable tbl, int nd_cap, int rows_pad, int dim, int ld,
                                                         const float* __restrict__ dn_all, __bf16* __restrict__ out_all)
{
    constexpr int KPB = K + 24;
    constexpr int NP = KPB / 8;                // 16-byte pieces per row
    const mif_match_pair& pr = tbl[blockIdx.y];
    const int nd = pr.d_nd ? min(*pr.d_nd, nd_cap) : nd_cap;
    const int rows = ((nd + kDC - 1) / kDC) * kDC;          // the chunks the search will read
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t / NP, p = t - j * NP;
    if (j >= rows) return;
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (__bf16)0.0f;
    if (p < K / 8) {
        if (j < nd) {
            const float* r = pr.d_db + (size_t)j * ld;
            float e[8];
            if (VEC && 8 * p + 8 <= ld) {
                const float4 u = *reinterpret_cast<const float4*>(r + 8 * p);
                const float4 w = *reinterpret_cast<const float4*>(r + 8 * p + 4);
                e[0] = u.x; e[1] = u.y; e[2] = u.z; e[3] = u.w; e[4] = w.x; e[5] = w.y; e[6] = w.z; e[7] = w.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) e[i] = (8 * p + i < dim) ? r[8 * p + i] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (__bf16)((8 * p + i < dim) ? e[i] : 0.0f);
        }
    } else if (p == K / 8) {
        const float hn = j < nd ? -0.5f * dn_all[(size_t)blockIdx.y * nd_cap + j] : kPadNorm;
        const __bf16 h1 = (__bf16)hn;
        const float r1 = hn - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        v[0] = h1; v[1] = h2; v[2] = (__bf16)(r1 - (float)h2);
    }
    *reinterpret_cast<bf16x8*>(out_all + ((size_t)blockIdx.y * rows_pad + j) * KPB + 8 * p) = v;
}

// The largest of three as the bare instruction. The compiler does not track the wait states between a
// matrix instruction and an assembly statement that reads its result: the first use of every accumulator
// must be an ordinary expression (max_raw), which this may then follow through a data dependence.
__device__ __forceinline__ float max3_raw(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

typedef const char __attribute__((address_space(1)))* GlobalBytes;
typedef char __attribute__((address_space(3)))* LdsBytes;

template <int K, bool VEC, int PASS>
__global__ __launch_bounds__(512, 1) void k_match_screen4(PairTable tbl, int npairs, int nq_cap, int nd_cap, int dim,
                                                          int ld, const float* __restrict__ qn_all,
                                                          const __bf16* __restrict__ dbf_all, int rows_pad,
                                                          const unsigned int* __restrict__ dnmax_all, int nsplit,
                                                          float* __restrict__ t2_all, size_t s1_off,
                                                          int* __restrict__ cnt_all, int2* __restrict__ cand_all,
                                                          unsigned int* __restrict__ work_counter, int sample)
{
    constexpr int KP = K + 16;                 // contraction length with the norm's k-step
    constexpr int KS = KP / 16;                // matrix instructions per tile
    constexpr int KPB = K + 24;                // row pitch in bf16 elements (16 bytes of padding)
    constexpr int WV = 8;
    constexpr int NT = 64 * WV;                // threads per block
    constexpr int kQB = kQW2 * WV;             // queries per block and work item
    constexpr int kChunkBytes = kDC * KPB * 2;
    constexpr int kPieces = kChunkBytes / 1024;             // wave-wide copies per chunk
    static_assert(kPieces * 1024 == kChunkBytes, "a chunk is a whole number of wave-wide copies");
    // one LDS array: the two chunk buffers, then the item tables
    constexpr int kOffFirst = 2 * kChunkBytes;
    constexpr int kOffRows = kOffFirst + (kMaxPairs + 1) * 4;
    constexpr int kOffItem = kOffRows + kMaxPairs * 2 * 4;
    __shared__ __attribute__((aligned(1024))) char s_all[kOffItem + 16];
    int* const s_first = reinterpret_cast<int*>(s_all + kOffFirst);
    int (*const s_rows)[2] = reinterpret_cast<int (*)[2]>(s_all + kOffRows);
    int* const s_item = reinterpret_cast<int*>(s_all + kOffItem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int col = lane & 31;
    const int group = wave >> 2;               // 0: matrix phase first, 1: half a chunk later

    for (int z = tid; z < kMaxPairs; z += NT) {
        int cnt = 0;
        if (z < npairs) {
            const mif_match_pair& pz = tbl[z];
            const int nqz = pz.d_nq ? min(*pz.d_nq, nq_cap) : nq_cap;
            const int ndz = pz.d_nd ? min(*pz.d_nd, nd_cap) : nd_cap;
            s_rows[z][0] = nqz;
            s_rows[z][1] = ndz;
            const int chunks = (ndz + kDC - 1) / kDC;
            const int per = (chunks + nsplit - 1) / nsplit;
            const int live = per > 0 ? (chunks + per - 1) / per : 0;         // splits that hold rows
            cnt = ((nqz + kQB - 1) / kQB) * live;
        }
        s_first[z + 1] = cnt;
    }
    __syncthreads();
    if (tid < 64) {
        constexpr int PER = kMaxPairs / 64;
        int c[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { c[j] = s_first[tid * PER + j + 1]; sum += c[j]; }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (tid >= d) incl += o;
        }
        int run = incl - sum;
        if (tid == 0) s_first[0] = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { run += c[j]; s_first[tid * PER + j + 1] = run; }
        if (tid == 0) *s_item = (int)atomicAdd(work_counter, 1u);
    }

  for (;;) {
    __syncthreads();
    const int item = __builtin_amdgcn_readfirstlane(*s_item);
    __syncthreads();
    if (item >= s_first[kMaxPairs]) break;     // uniform over the block
    int next_item = 0;
    if (tid == 0) next_item = (int)atomicAdd(work_counter, 1u);
    int z = 0;
    {
        int hi = npairs - 1;
        while (z < hi) {
            const int mid = (z + hi + 1) >> 1;
            if (s_first[mid] <= item) z = mid; else hi = mid - 1;
        }
    }
    z = __builtin_amdgcn_readfirstlane(z);
    const mif_match_pair& pr = tbl[z];
    const float* __restrict__ Q = pr.d_q;
    const float* __restrict__ qn = qn_all + (size_t)z * nq_cap;
    const int nq = __builtin_amdgcn_readfirstlane(s_rows[z][0]);
    const int nd = __builtin_amdgcn_readfirstlane(s_rows[z][1]);
    const int qtiles = (nq + kQB - 1) / kQB;
    const int split = (item - s_first[z]) / qtiles;             // consecutive items share database rows
    const int qbase = ((item - s_first[z]) - split * qtiles) * kQB + wave * kQW2;
    const int chunks = (nd + kDC - 1) / kDC;
    const int per = (chunks + nsplit - 1) / nsplit;
    const int c_lo = split * per;
    const int c_hi = min(chunks, c_lo + per);
    float* __restrict__ t2s = t2_all + (size_t)z * nsplit * nq_cap;
    int* __restrict__ cnt = cnt_all + ((size_t)z * nsplit + split) * nq_cap;
    int2* __restrict__ cand = cand_all + ((size_t)z * nsplit + split) * nq_cap * kCandCap;
    // the pair's bf16 image; a chunk is kChunkBytes contiguous bytes of it
    const GlobalBytes image = (GlobalBytes)(dbf_all + (size_t)z * rows_pad * KPB);
    // chunk c into buffer buf: this wave's share of the wave-wide copies, pieces first, first + step, ...
    auto copy_chunk = [&](int c, int buf, int first, int step) {
        const GlobalBytes src = image + (size_t)c * kChunkBytes + lane * 16;
#pragma unroll
        for (int p0 = 0; p0 < kPieces + 8; p0 += 4) {       // (step is 4 or 8: at most kPieces / 4 + 1 pieces per wave)
            const int p = first + (p0 >> 2) * step;
            if (p < kPieces)
                __builtin_amdgcn_global_load_lds(src + p * 1024, (LdsBytes)(s_all + buf * kChunkBytes + p * 1024), 16, 0, 0);
        }
    };
    copy_chunk(c_lo, 0, wave, WV);

    // queries (B operands): lane holds q[qbase + 32 s + col][16 j + 8 half .. + 7] for every instruction j
    bf16x8 qf[2][KS];
    float qnv[2];
    int qi[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
        qi[s2] = qbase + 32 * s2 + col;
        const float* qr = Q + (size_t)min(qi[s2], nq - 1) * ld;
        const bool row_ok = qi[s2] < nq;
#pragma unroll
        for (int j = 0; j < KS - 1; j++) {
            const int k0 = 16 * j + 8 * half;
            float e[8];
            if (VEC) {
                const float4 u = *reinterpret_cast<const float4*>(qr + min(k0, (dim - 1) & ~3));
                const float4 w = *reinterpret_cast<const float4*>(qr + min(k0 + 4, (dim - 1) & ~3));
                e[0] = u.x; e[1] = u.y; e[2] = u.z; e[3] = u.w; e[4] = w.x; e[5] = w.y; e[6] = w.z; e[7] = w.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) e[i] = qr[min(k0 + i, dim - 1)];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) qf[s2][j][i] = (__bf16)((row_ok && k0 + i < dim) ? e[i] : 0.0f);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) qf[s2][KS - 1][i] = (__bf16)((row_ok && half == 0 && i < 3) ? 1.0f : 0.0f);
        qnv[s2] = row_ok ? qn[qi[s2]] : 0.0f;
    }
    // pass 0: the largest a seen among the rows of the chunks' first row tiles and among those of their
    // second row tiles. The smaller of the two maxima belongs to a row other than the overall best one, so
    // as a score it is at least the second smallest score: a threshold as valid as the second smallest
    // itself (a little looser: it is the second smallest whenever the two best rows sit in different
    // tiles) for a quarter of the vector instructions (one v_max3 per two values).
    float t1[2] = {-kInf, -kInf}, t2[2] = {-kInf, -kInf};
    // pass 1: a row is kept when its score |q|^2 - 2 a is at most the query's threshold, i.e. when
    // a >= (|q|^2 - threshold) / 2. The second smallest score over all rows is at most the second
    // smallest of any one split, so the smallest of the splits' values (+ 2 m) keeps every winner.
    float alim[2] = {kInf, kInf};
    int filled[2] = {0, 0};
    if (PASS == 1) {            // (k_match_limits left the query's limit in the slot of split 0)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
            if (qi[s2] < nq) alim[s2] = t2s[qi[s2]];
    }

    auto consider = [&](const f32x16& acc, int jbase, int s2, bool full) {
        if (PASS == 0) {
            float& t = (jbase & 32) ? t2[s2] : t1[s2];
            t = max_raw(t, acc[0]);
#pragma unroll
            for (int r = 1; r < 15; r += 2) t = max3_raw(t, acc[r], acc[r + 1]);
            t = max_raw(t, acc[15]);
        } else {
            float mx = max_raw(acc[0], acc[1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) mx = max3_raw(mx, acc[r], acc[r + 1]);
            // A query's rows are seen by the two lanes l, l + 32 of this wave only, so its list is
            // filled without atomics: the branch is uniform over the wave, the lower lane's rows of
            // a tile go first, and both lanes keep the same running count.
            if (__builtin_amdgcn_ballot_w64(mx >= alim[s2]) != 0ull) {
                unsigned int hits = 0u;
#pragma unroll
                for (int r = 0; r < 16; r++) hits |= (acc[r] >= alim[s2] ? 1u : 0u) << r;
                // What goes into the list with a row: the largest product of this lane's 16 rows, which is the
                // product of the best row kept here and an upper bound for the others (k_match_rank tightens
                // the threshold with these before it reads any row).
                float top = mx;
                if (!full) {                    // rows staged beyond the database end (they pass an infinite threshold)
                    top = -kInf;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        if (jbase + (r & 3) + 8 * (r >> 2) + 4 * half >= nd) hits &= ~(1u << r);
                        else top = fmaxf(top, acc[r]);
                    }
                }
                const int topbits = __float_as_int(top);
                const int mine = __popc(hits);
                const int other = __shfl_xor(mine, 32);
                int at = filled[s2] + (half ? other : 0);
                filled[s2] += mine + other;
                while (hits != 0u) {
                    const int r = __ffs((int)hits) - 1;
                    hits &= hits - 1u;
                    if (at < kCandCap)
                        cand[(size_t)qi[s2] * kCandCap + at] = make_int2(jbase + (r & 3) + 8 * (r >> 2) + 4 * half, topbits);
                    at++;
                }
            }
        }
    };
    f32x16 acc00, acc01, acc10, acc11;          // [row tile][query set] of the chunk last multiplied
    auto matrix_phase = [&](int buf) {
#pragma unroll
        for (int r = 0; r < 16; r++) { acc00[r] = 0.0f; acc01[r] = 0.0f; acc10[r] = 0.0f; acc11[r] = 0.0f; }
        const __bf16* ap0 = reinterpret_cast<const __bf16*>(s_all + buf * kChunkBytes) + col * KPB + 8 * half;
        const __bf16* ap1 = ap0 + 32 * KPB;
        // the A operands are read from LDS two k-steps ahead of the matrix instructions that use them
        // (one step ahead, 128 cycles, did not cover the LDS latency under load: the matrix pipe was busy
        // for 56 % of a matrix phase)
        bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ap0);
        bf16x8 b0 = *reinterpret_cast<const bf16x8*>(ap1);
        bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ap0 + 16);
        bf16x8 b1 = *reinterpret_cast<const bf16x8*>(ap1 + 16);
        // the order the scheduler must keep: the four reads above, then per k-step its four matrix
        // instructions followed by the two reads for two steps later
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int j = 0; j < KS; j++) {
            bf16x8 a2 = a1, b2 = b1;
            if (j + 2 < KS) {
                a2 = *reinterpret_cast<const bf16x8*>(ap0 + 16 * (j + 2));
                b2 = *reinterpret_cast<const bf16x8*>(ap1 + 16 * (j + 2));
            }
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qf[0][j], acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qf[1][j], acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, qf[0][j], acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, qf[1][j], acc11, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (j + 2 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            a0 = a1; b0 = b1; a1 = a2; b1 = b2;
        }
    };
    auto vector_phase = [&](int c) {            // epilogue of chunk c
        const bool full = (c + 1) * kDC <= nd;
        consider(acc00, c * kDC, 0, full);
        consider(acc01, c * kDC, 1, full);
        consider(acc10, c * kDC + 32, 0, full);
        consider(acc11, c * kDC + 32, 1, full);
    };
    // chunk c_lo has to be in LDS before the first matrix phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // pass 0 may look at every `sample`-th chunk only: the bounds it leaves then hold for a subset of the
    // rows, and the second smallest score of a subset is at least that of all rows (a looser, still
    // valid limit for pass 1, for an n-th of the matrix work)
    const int cs = PASS == 0 ? sample : 1;
    const int c_last = c_lo + (c_hi - 1 - c_lo) / cs * cs;
    int buf = 0;
    for (int c = c_lo; c < c_hi; c += cs, buf ^= 1) {
        // The next chunk is fetched by the half that is in its (shorter) vector phase: issuing a copy
        // costs a wave 60-180 cycles, which the matrix phase cannot spare. Its buffer was last read in
        // half period 2c - 1.
        if (group == 0) {
            matrix_phase(buf);
        } else {
            if (c + cs < c_hi) copy_chunk(c + cs, buf ^ 1, wave - 4, 4);
            if (c > c_lo) vector_phase(c - cs);
        }
        __builtin_amdgcn_s_barrier();
        if (group == 0) vector_phase(c);
        else matrix_phase(buf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the pieces of chunk c + 1 this wave issued have landed
        __builtin_amdgcn_s_barrier();
    }
    if (group == 1) vector_phase(c_last);
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
        if (PASS == 0) {
            // lanes l and l + 32 hold the same query (interleaved database rows)
            // four disjoint classes of rows (this lane's or the other's, even or odd tiles): the largest
            // of their maxima is the split's largest result and the second largest of them a bound on
            // its second largest
            const float e0 = t1[s2], o0 = t2[s2], e1 = __shfl_xor(t1[s2], 32), o1 = __shfl_xor(t2[s2], 32);
            const float ea = fmaxf(e0, e1), eb = fminf(e0, e1), oa = fmaxf(o0, o1), ob = fminf(o0, o1);
            const float a1 = fmaxf(ea, oa);
            const float a2 = fmaxf(fminf(ea, oa), fmaxf(eb, ob));
            // as scores; +inf for a class without rows (-inf or a padded row's -3e38)
            if (half == 0 && qi[s2] < nq) {
                t2s[(size_t)split * nq_cap + qi[s2]] = __builtin_fmaf(-2.0f, a2, qnv[s2]);
                t2s[s1_off + (size_t)split * nq_cap + qi[s2]] = __builtin_fmaf(-2.0f, a1, qnv[s2]);
            }
        } else if (half == 0 && qi[s2] < nq) {
            cnt[qi[s2]] = filled[s2];
        }
    }
    if (tid == 0) *s_item = next_item;
  }
}

// A query whose candidate lists do not settle its answer: the splits whose rows all have to be ranked
// (bit sp of `splits`), for k_match_overflow.
struct OverflowEntry {
    int pair, query;
    unsigned long long splits;
};

// Ranks the candidates of the screening passes with the exact search's arithmetic and writes the
// two winners where k_match_finalize reads them (slot 0 of the query; slot 1 is for k_match_overflow
// and starts empty). A list that overflowed stands for all rows of its split, and lists that hold fewer
// rows than the answer has (scores that are not numbers compare false) are not trusted at all: such
// queries go to the overflow queue (one entry per query at most, so the queue of pairs * nq_cap
// entries cannot fill) and a block of k_match_overflow ranks the rows of those splits.
// grid (ceil(nq_cap/128), npairs), one thread per query.
template <bool VEC>
__global__ __launch_bounds__(128) void k_match_rank(PairTable tbl, int nq_cap, int nd_cap, int dim, int ld,
                                                    const float* __restrict__ qn_all, const float* __restrict__ dn_all,
                                                    int nsplit, const int* __restrict__ cnt_all,
                                                    const int2* __restrict__ cand_all,
                                                    float* __restrict__ part_s_all, int* __restrict__ part_i_all,
                                                    OverflowEntry* __restrict__ queue, unsigned int* __restrict__ queue_count,
                                                    const unsigned int* __restrict__ dnmax_all, int refilter)
{
    const mif_match_pair& pr = tbl[blockIdx.y];
    const int nq = pr.d_nq ? min(*pr.d_nq, nq_cap) : nq_cap;
    const int nd = pr.d_nd ? min(*pr.d_nd, nd_cap) : nd_cap;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float* __restrict__ Q = pr.d_q;
    const float* __restrict__ D = pr.d_db;
    const float qnv = qn_all[(size_t)blockIdx.y * nq_cap + i];
    const float* __restrict__ dn = dn_all + (size_t)blockIdx.y * nd_cap;
    const int chunks = (nd + kDC - 1) / kDC;
    const int per = (chunks + nsplit - 1) / nsplit;
    Top2 t;
    t.s1 = t.s2 = kInf;
    t.i1 = t.i2 = 0x7fffffff;
    const float* q = Q + (size_t)i * ld;
    // the query's row stays in registers (dim <= 128): every candidate row is met by the same values
    float vq[4][32];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int k0 = 32 * b;
        if (k0 < dim) load_span<VEC>(q + k0, min(32, ld - k0), vq[b]);
    }
    int listed = 0;
    unsigned long long deferred = 0ull, with_rows = 0ull;
    // After a sampled first pass (refilter): the lists were filled against the looser threshold such a pass
    // leaves and hold far more rows than the answer needs (about 160 per query at 1 M x 1 M rows with every
    // 8th chunk sampled, spread over 32 lists). Every entry carries a bf16 product a (score s~ = |q|^2 - 2 a):
    // the largest of the 16 rows its lane looked at in that tile, i.e. the product of the best row the lane
    // kept there and an upper bound for the others it kept (they follow it in the list with the same
    // value). The distinct values are therefore the products of distinct listed rows; their second largest
    // a2 gives the second smallest s~ of a subset of the rows, a bound as valid as pass 0's and tighter,
    // and only the rows whose value is >= a2 - m (the margin of k_match_limits) are read and ranked: every
    // row among the two nearest has a >= a2 - m, and its entry's value is at least its a. About half of
    // the listed rows go (the margin m is a worst-case bound and wide against the spread of the best
    // scores). Lists that overflowed are left to k_match_overflow as before.
    float alim = -kInf;
    if (refilter) {
        float a1 = -kInf, a2 = -kInf;
        int prev = 0x7fc00000;           // (a NaN: no product has these bits)
        for (int sp = 0; sp * per < chunks; sp++) {
            const size_t at = ((size_t)blockIdx.y * nsplit + sp) * nq_cap + i;
            const int n = cnt_all[at];
            if (n > kCandCap) continue;
            const int2* __restrict__ list = cand_all + at * kCandCap;
            for (int c = 0; c < n; c++) {
                const int bits = list[c].y;
                if (bits == prev) continue;     // the same lane's rows of one tile share their value
                prev = bits;
                const float a = __int_as_float(bits);
                a2 = fmaxf(a2, fminf(a1, a));
                a1 = fmaxf(a1, a);
            }
        }
        const float dmax = __builtin_bit_cast(float, dnmax_all[blockIdx.y]);
        const float m = kScreenRel * __builtin_sqrtf(qnv * dmax) + 3.0f * kScreenAbs * (qnv + dmax);
        alim = a2 - m;                  // -inf while fewer than two rows are listed: every listed row is ranked
    }
    for (int sp = 0; sp * per < chunks; sp++) {
        const size_t at = ((size_t)blockIdx.y * nsplit + sp) * nq_cap + i;
        const int n = cnt_all[at];
        listed += n;
        with_rows |= 1ull << sp;
        if (n > kCandCap) {
            deferred |= 1ull << sp;
            continue;
        }
        const int2* __restrict__ list = cand_all + at * kCandCap;
        for (int c = 0; c < n; c++) {
            const int2 e = list[c];
            if (refilter && !(__int_as_float(e.y) >= alim)) continue;
            const int j = e.x;
            const float* r = D + (size_t)j * ld;
            float acc = 0.0f;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int k0 = 32 * b;
                if (k0 < dim) {
                    const int span = min(32, dim - k0);
                    float vd[32];
                    load_span<VEC>(r + k0, min(32, ld - k0), vd);
#pragma unroll
                    for (int k = 0; k < 32; k++)
                        if (k < span) acc = __builtin_fmaf(vd[k], vq[b][k], acc);
                }
            }
            top2_insert(t, __builtin_fmaf(-2.0f, acc, qnv + dn[j]), j);
        }
    }
    if (listed < min(2, nd)) {          // the lists cannot hold the answer: every row is ranked
        deferred = with_rows;
        t.s1 = t.s2 = kInf;
        t.i1 = t.i2 = 0x7fffffff;
    }
    if (deferred != 0ull) {
        const unsigned int e = atomicAdd(queue_count, 1u);
        queue[e].pair = blockIdx.y;
        queue[e].query = i;
        queue[e].splits = deferred;
    }
    // slots [pair][2][nq_cap][2]
    const size_t o = ((size_t)blockIdx.y * 2 * nq_cap + i) * 2;
    part_s_all[o] = t.s1; part_s_all[o + 1] = t.s2;
    part_i_all[o] = t.i1; part_i_all[o + 1] = t.i2;
    const size_t o1 = o + (size_t)nq_cap * 2;
    part_s_all[o1] = kInf; part_s_all[o1 + 1] = kInf;
    part_i_all[o1] = 0x7fffffff; part_i_all[o1 + 1] = 0x7fffffff;
}

// The queries k_match_rank queued: a block per entry (persistent blocks stride over the queue) ranks all
// rows of the entry's splits, a row per thread and step, and leaves the two winners in slot 1 of the
// query. The order in which rows meet does not matter: winners are chosen on (score, index).
template <bool VEC>
__global__ __launch_bounds__(256) void k_match_overflow(PairTable tbl, int nq_cap, int nd_cap, int dim, int ld,
                                                        const float* __restrict__ qn_all, const float* __restrict__ dn_all,
                                                        int nsplit, const OverflowEntry* __restrict__ queue,
                                                        const unsigned int* __restrict__ queue_count,
                                                        float* __restrict__ part_s_all, int* __restrict__ part_i_all)
{
    __shared__ Top2 s_top[256];
    const unsigned int n_entries = *queue_count;
    const int tid = threadIdx.x;
    for (unsigned int e = blockIdx.x; e < n_entries; e += gridDim.x) {      // uniform over the block
        const OverflowEntry en = queue[e];
        const mif_match_pair& pr = tbl[en.pair];
        const int nd = pr.d_nd ? min(*pr.d_nd, nd_cap) : nd_cap;
        const float* __restrict__ D = pr.d_db;
        const float* q = pr.d_q + (size_t)en.query * ld;
        const float qnv = qn_all[(size_t)en.pair * nq_cap + en.query];
        const float* __restrict__ dn = dn_all + (size_t)en.pair * nd_cap;
        const int chunks = (nd + kDC - 1) / kDC;
        const int per = (chunks + nsplit - 1) / nsplit;
        float vq[4][32];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int k0 = 32 * b;
            if (k0 < dim) load_span<VEC>(q + k0, min(32, ld - k0), vq[b]);
        }
        Top2 t;
        t.s1 = t.s2 = kInf;
        t.i1 = t.i2 = 0x7fffffff;
        unsigned long long splits = en.splits;
        while (splits != 0ull) {
            const int sp = __ffsll((long long)splits) - 1;
            splits &= splits - 1ull;
            const int lo = sp * per * kDC;
            const int hi = min(nd, lo + per * kDC);
            for (int j = lo + tid; j < hi; j += 256) {
                const float* r = D + (size_t)j * ld;
                float acc = 0.0f;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int k0 = 32 * b;
                    if (k0 < dim) {
                        const int span = min(32, dim - k0);
                        float vd[32];
                        load_span<VEC>(r + k0, min(32, ld - k0), vd);
#pragma unroll
                        for (int k = 0; k < 32; k++)
                            if (k < span) acc = __builtin_fmaf(vd[k], vq[b][k], acc);
                    }
                }
                top2_insert(t, __builtin_fmaf(-2.0f, acc, qnv + dn[j]), j);
            }
        }
        s_top[tid] = t;
        __syncthreads();
        for (int stride = 128; stride > 0; stride >>= 1) {
            if (tid < stride) {
                Top2 a = s_top[tid];
                const Top2 o = s_top[tid + stride];
                top2_insert(a, o.s1, o.i1);
                top2_insert(a, o.s2, o.i2);
                s_top[tid] = a;
            }
            __syncthreads();
        }
        if (tid == 0) {
            const Top2 w = s_top[0];
            const size_t o1 = (((size_t)en.pair * 2 + 1) * nq_cap + en.query) * 2;
            part_s_all[o1] = w.s1; part_s_all[o1 + 1] = w.s2;
            part_i_all[o1] = w.i1; part_i_all[o1 + 1] = w.i2;
        }
        __syncthreads();                    // s_top is reused by the next entry
    }
}

// grid (ceil(nq_cap/128), npairs). A thread owns a query: it merges the splits'
// candidates, then evaluates the exact distances of both winners together, one
// span of 32 elements at a time (all loads of a span in flight, then the two
// sequential fmaf chains).
template <bool VEC>
__global__ __launch_bounds__(128) void k_match_finalize(PairTable tbl, int nq_cap, int nd_cap, int dim, int ld,
                                                        int nsplit, const float* __restrict__ part_s,
                                                        const int* __restrict__ part_i)
{
    const mif_match_pair& pr = tbl[blockIdx.y];
    const float* __restrict__ Q = pr.d_q;
    const float* __restrict__ D = pr.d_db;
    const int* d_nq = pr.d_nq;
    const int* d_nd = pr.d_nd;
    int* __restrict__ idx2 = pr.d_idx2;
    float* __restrict__ dist2 = pr.d_dist2;
    // nsplit < 0: -nsplit result slots per query that are all filled (the screened search: slot 0 from
    // k_match_rank, slot 1 from k_match_overflow); otherwise one slot per database split
    const bool all_slots = nsplit < 0;
    if (all_slots) nsplit = -nsplit;
    part_s += (size_t)blockIdx.y * nsplit * nq_cap * 2;
    part_i += (size_t)blockIdx.y * nsplit * nq_cap * 2;
    const int nq = d_nq ? min(*d_nq, nq_cap) : nq_cap;
    const int nd = d_nd ? min(*d_nd, nd_cap) : nd_cap;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    Top2 t;
    t.s1 = t.s2 = kInf;
    t.i1 = t.i2 = 0x7fffffff;
    const int chunks = (nd + kDC - 1) / kDC;
    const int per = (chunks + nsplit - 1) / nsplit;
    for (int sp = 0; sp < nsplit; sp++) {
        if (!all_slots && sp * per >= chunks) break;    // this split had no database rows
        const size_t o = ((size_t)sp * nq_cap + i) * 2;
        const float2 ps = *reinterpret_cast<const float2*>(&part_s[o]);
        const int2 pi = *reinterpret_cast<const int2*>(&part_i[o]);
        top2_insert(t, ps.x, pi.x);
        top2_insert(t, ps.y, pi.y);
    }
    int j1 = (t.i1 < nd) ? t.i1 : -1;
    int j2 = (t.i2 < nd) ? t.i2 : -1;
    const float* q = Q + (size_t)i * ld;
    // a missing winner is evaluated against row 0 (or the query itself) and discarded
    const float* r1 = (j1 >= 0) ? D + (size_t)j1 * ld : q;
    const float* r2 = (j2 >= 0) ? D + (size_t)j2 * ld : q;
    float a1 = 0.0f, a2 = 0.0f;
    for (int k0 = 0; k0 < dim; k0 += 32) {
        const int span = min(32, dim - k0);
        const int avail = min(32, ld - k0);
        float vq[32], v1[32], v2[32];
        load_span<VEC>(q + k0, avail, vq);
        load_span<VEC>(r1 + k0, avail, v1);
        load_span<VEC>(r2 + k0, avail, v2);
#pragma unroll
        for (int k = 0; k < 32; k++) {
            if (k < span) {
                const float d1 = vq[k] - v1[k];
                const float d2 = vq[k] - v2[k];
                a1 = __builtin_fmaf(d1, d1, a1);
                a2 = __builtin_fmaf(d2, d2, a2);
            }
        }
    }
    float e1 = (j1 >= 0) ? __builtin_sqrtf(a1) : kInf;
    float e2 = (j2 >= 0) ? __builtin_sqrtf(a2) : kInf;
    float sc1 = (j1 >= 0) ? t.s1 : kInf, sc2 = (j2 >= 0) ? t.s2 : kInf;
    if (j2 >= 0 && (e2 < e1 || (e2 == e1 && j2 < j1))) {
        const float te = e1; e1 = e2; e2 = te;
        const int tj = j1; j1 = j2; j2 = tj;
        const float ts = sc1; sc1 = sc2; sc2 = ts;
    }
    *reinterpret_cast<int2*>(&idx2[2 * i]) = make_int2(j1, j2);
    *reinterpret_cast<float2*>(&dist2[2 * i]) = make_float2(e1, e2);
    if (pr.d_score2) *reinterpret_cast<float2*>(&pr.d_score2[2 * i]) = make_float2(sc1, sc2);
}

// Good-match filters of the reference on the device, one block per pair.
//   mode 1 (SIFT overload, kpmatch.h:177-202): min_dist starts at 100, max_dist at 0, both
//     follow the nearest distances; keep dist <= max(2.5 * min_dist, 0.35), all in double.
//   mode 2 (SURF overload, kpmatch.h:289-297): keep d0 < 0.75f * d1.
// good[i] = 1/0 per query row; summary = {kept, query rows}; minmax = {min_dist, max_dist}.
__global__ __launch_bounds__(256) void k_match_filter(PairTable tbl, int nq_cap, int mode)
{
    __shared__ double s_min[256], s_max[256];
    __shared__ int s_cnt[256];
    const mif_match_pair& pr = tbl[blockIdx.x];
    if (!pr.d_good || !pr.d_summary) return;
    const int nq = pr.d_nq ? min(*pr.d_nq, nq_cap) : nq_cap;
    const float* __restrict__ dist2 = pr.d_dist2;
    const int tid = threadIdx.x;
    double mn = 100.0, mx = 0.0;
    if (mode == 1) {
        for (int i = tid; i < nq; i += 256) {
            const double d = (double)dist2[2 * i];
            if (d < mn) mn = d;
            if (d > mx) mx = d;
        }
        s_min[tid] = mn; s_max[tid] = mx;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) {
                if (s_min[tid + st] < s_min[tid]) s_min[tid] = s_min[tid + st];
                if (s_max[tid + st] > s_max[tid]) s_max[tid] = s_max[tid + st];
            }
            __syncthreads();
        }
        mn = s_min[0]; mx = s_max[0];
    }
    const double lim1 = 2.5 * mn;
    const double thr = lim1 > 0.35 ? lim1 : 0.35;
    int cnt = 0;
    for (int i = tid; i < nq; i += 256) {
        bool keep;
        if (mode == 1) keep = (double)dist2[2 * i] <= thr;
        else keep = dist2[2 * i] < 0.75f * dist2[2 * i + 1];
        pr.d_good[i] = keep ? 1 : 0;
        cnt += keep ? 1 : 0;
    }
    s_cnt[tid] = cnt;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) s_cnt[tid] += s_cnt[tid + st];
        __syncthreads();
    }
    if (tid == 0) {
        pr.d_summary[0] = s_cnt[0];
        pr.d_summary[1] = nq;
        if (pr.d_minmax) { pr.d_minmax[0] = (float)mn; pr.d_minmax[1] = (float)mx; }
    }
}

} // namespace

struct mif_matcher {
    int device = 0, max_q = 0, max_d = 0, max_pairs = 1;
    float* qn = nullptr;       // [pairs][max_q]
    float* dn = nullptr;       // [pairs][max_d]
    float* part_s = nullptr;   // [pairs][kMaxSplit][max_q][2]
    int* part_i = nullptr;
    // staging for the host-buffer entry point
    float* d_q = nullptr;
    float* d_db = nullptr;
    size_t d_q_bytes = 0, d_db_bytes = 0;
    int* d_idx = nullptr;
    float* d_dist = nullptr;
    float* d_score = nullptr;
    unsigned int* work = nullptr;   // counters: [0] exact search, [1], [2] screening passes, [3] queued queries
    OverflowEntry* ovf = nullptr;   // [pairs][max_q] queries whose lists do not settle their answer
    double last_passes = 0.0;       // contraction passes over the database the last call's screening made
    // screening: largest database norm per pair; second smallest score / candidate count / candidates
    // per (pair, split, query), `screen_slots` of them
    unsigned int* dnmax = nullptr;  // [pairs]
    float* thr = nullptr;
    int* cand_cnt = nullptr;
    int2* cand = nullptr;           // [screen_slots][kCandCap] (row, bits of the bf16 product it was kept for)
    size_t screen_slots = 0;
    // bf16 image of the database rows for the screening passes (k_match_pack_rows): [pairs][rows_pad][152]
    __bf16* dbf = nullptr;
    int rows_pad = 0;
    // pair tables: pinned host copies and their device twins, used in rotation; a slot is reused
    // once the call that read it last has finished on the GPU (slot_ev)
    mif_match_pair* h_tbl[kTableSlots] = {};
    mif_match_pair* d_tbl[kTableSlots] = {};
    hipEvent_t slot_ev[kTableSlots] = {};
    bool slot_used[kTableSlots] = {};
    int next_slot = 0;
    int blocks = 512;               // persistent blocks: two per compute unit
    bool screening = true;          // mif_match_set_screening
};

namespace {

constexpr int kMaxSplit = 64;
// Calls with at least this many scores to evaluate go through the bf16 screening passes; below,
// the two extra launches cost more than the cheaper arithmetic saves.
constexpr double kScreenMinScores = 3.2e7;
constexpr int kScreenSplitRows = 32768;     // most rows of a database split under screening
// The screening passes through the bf16 image (k_match_pack_rows, k_match_screen4) pay for the image and for
// their long start of a work item only on large searches: measured against k_match_screen2 they lose at
// 8192 x 8192 rows and below, draw at 16384 x 65536 and win 9-17 % from 65536 x 65536 to 1 M x 1 M. Row counts
// known on the host only.
constexpr double kTurnsMinScores = 2e9;
constexpr int kTurnsMinQueries = 16384;
constexpr int kTurnsMinRows = 65536;

void free_matcher(mif_matcher* m)
{
    (void)hipFree(m->qn); (void)hipFree(m->dn); (void)hipFree(m->part_s); (void)hipFree(m->part_i);
    (void)hipFree(m->d_q); (void)hipFree(m->d_db); (void)hipFree(m->d_idx); (void)hipFree(m->d_dist);
    (void)hipFree(m->d_score); (void)hipFree(m->work);
    (void)hipFree(m->dnmax); (void)hipFree(m->thr); (void)hipFree(m->cand_cnt); (void)hipFree(m->cand);
    (void)hipFree(m->dbf); (void)hipFree(m->ovf);
    for (int i = 0; i < kTableSlots; i++) {
        if (m->h_tbl[i]) (void)hipHostFree(m->h_tbl[i]);
        (void)hipFree(m->d_tbl[i]);
        if (m->slot_ev[i]) (void)hipEventDestroy(m->slot_ev[i]);
    }
}

int pick_split(int nq, int nd, int npairs, bool device_counts)
{
    // Enough blocks to load every CU several times over, but no split shorter than 4 chunks.
    // With device-side row counts nq / nd are capacities; the rows present are assumed to be
    // an eighth of them (blocks beyond the real counts exit at once).
    if (const char* e = getenv("MIF_MATCH_SPLIT"))       // tuning override
        return std::max(1, std::min(atoi(e), kMaxSplit));
    if (device_counts) {
        nq = std::max(1, nq / 8);
        nd = std::max(1, nd / 8);
    }
    const int qblocks = npairs * ((nq + kQT * kWaves - 1) / (kQT * kWaves));
    const int chunks = (nd + kDC - 1) / kDC;
    int split = (4096 + qblocks - 1) / qblocks;
    split = std::min(split, std::max(1, chunks / 4));
    return std::max(1, std::min(split, kMaxSplit));
}

int match_batch(mif_matcher* m, const mif_match_pair* pairs, int npairs, int nq, int nd, int dim, int ld,
                int filter_mode, hipStream_t st)
{
    if (npairs < 0 || npairs > m->max_pairs) {
        mif::set_error("match: %d pairs exceed the matcher's capacity of %d", npairs, m->max_pairs);
        return MIF_ERR_CAPACITY;
    }
    if (nq < 0 || nd < 0 || nq > m->max_q || nd > m->max_d) {
        mif::set_error("match: %d x %d rows exceed the matcher's capacity %d x %d", nq, nd, m->max_q, m->max_d);
        return MIF_ERR_CAPACITY;
    }
    if (dim < 1 || dim > 128 || ld < dim) {
        mif::set_error("match: dim must be 1..128 and ld >= dim (got dim %d, ld %d)", dim, ld);
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if (filter_mode < 0 || filter_mode > 2) {
        mif::set_error("match: unknown filter mode %d", filter_mode);
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if (nq == 0 || npairs == 0) return MIF_OK;
    for (int i = 0; i < npairs; i++) {
        if (!pairs[i].d_q || !pairs[i].d_idx2 || !pairs[i].d_dist2 || (nd > 0 && !pairs[i].d_db)) {
            mif::set_error("match: pair %d has a NULL pointer", i);
            return MIF_ERR_INVALID_ARGUMENT;
        }
    }
    // the call's pair table goes to the device through the next slot of the rotation
    const int slot = m->next_slot;
    m->next_slot = (slot + 1) % kTableSlots;
    if (m->slot_used[slot]) MIF_HIP(hipEventSynchronize(m->slot_ev[slot]));
    memcpy(m->h_tbl[slot], pairs, (size_t)npairs * sizeof(mif_match_pair));
    MIF_HIP(hipMemcpyAsync(m->d_tbl[slot], m->h_tbl[slot], (size_t)npairs * sizeof(mif_match_pair),
                           hipMemcpyHostToDevice, st));
    m->slot_used[slot] = true;
    for (int base = 0; base < npairs; base += kMaxPairs) {
        const int np = std::min(kMaxPairs, npairs - base);
        const mif_match_pair* hp = pairs + base;
        const mif_match_pair* tbl = m->d_tbl[slot] + base;
        float* qn = m->qn + (size_t)base * m->max_q;
        float* dn = m->dn + (size_t)base * m->max_d;
        float* ps = m->part_s + (size_t)base * kMaxSplit * m->max_q * 2;
        int* pi = m->part_i + (size_t)base * kMaxSplit * m->max_q * 2;
        bool device_counts = false;
        for (int i = 0; i < np; i++) device_counts = device_counts || hp[i].d_nq || hp[i].d_nd;
        int nsplit = pick_split(nq, nd, np, device_counts);
        // 16-byte loads of rows need ld % 4 == 0 and 16-byte aligned row starts
        bool vec_db = (ld % 4) == 0, vec_q = (ld % 4) == 0;
        for (int i = 0; i < np; i++) {
            vec_db = vec_db && ((reinterpret_cast<uintptr_t>(hp[i].d_db) & 15u) == 0);
            vec_q = vec_q && ((reinterpret_cast<uintptr_t>(hp[i].d_q) & 15u) == 0);
        }
        static const bool no_screen = getenv("MIF_MATCH_NO_SCREEN") != nullptr;      // A/B switch
        // Screening (k_match_screen): worth its two extra launches from kScreenMinScores scores on (with
        // device-side row counts the rows present are taken as an eighth of the capacities, as
        // pick_split does). A candidate list that overflows makes one thread rank all rows of the
        // list's split, so a split is kept to kScreenSplitRows rows: large databases get more splits
        // (which also spreads a query's candidates over more lists), and a database that would still
        // exceed that is searched exactly.
        bool screen = !no_screen && m->screening && nd >= 2 * kDC &&
                      (double)np * nq * nd * (device_counts ? 1.0 / 64.0 : 1.0) >= kScreenMinScores;
        if (screen) {
            const int chunks = (nd + kDC - 1) / kDC;
            int want = nsplit;
            if (!device_counts) want = std::max(want, (chunks * kDC + kScreenSplitRows / 2 - 1) / (kScreenSplitRows / 2));
            want = std::min(want, kMaxSplit);
            const int per = (chunks + want - 1) / want;
            if ((size_t)per * kDC <= (size_t)kScreenSplitRows &&
                (size_t)np * want * nq <= m->screen_slots - (size_t)base * m->max_q)
                nsplit = want;
            else
                screen = false;
        }
        m->last_passes = 0.0;
        if (nd > 0) {
            unsigned int* nmax = screen ? m->dnmax + base : nullptr;
            if (screen) MIF_HIP(hipMemsetAsync(nmax, 0, (size_t)np * sizeof(unsigned int), st));
            if (vec_db)
                hipLaunchKernelGGL(k_row_norms<true>, dim3((nd + 255) / 256, np), dim3(256), 0, st, tbl, 1, nd, dim, ld, dn, nmax);
            else
                hipLaunchKernelGGL(k_row_norms<false>, dim3((nd + 255) / 256, np), dim3(256), 0, st, tbl, 1, nd, dim, ld, dn, nmax);
            if (vec_q)
                hipLaunchKernelGGL(k_row_norms<true>, dim3((nq + 255) / 256, np), dim3(256), 0, st, tbl, 0, nq, dim, ld, qn, (unsigned int*)nullptr);
            else
                hipLaunchKernelGGL(k_row_norms<false>, dim3((nq + 255) / 256, np), dim3(256), 0, st, tbl, 0, nq, dim, ld, qn, (unsigned int*)nullptr);
            MIF_HIP(hipMemsetAsync(m->work, 0, 4 * sizeof(unsigned int), st));
            const bool vec = vec_db && vec_q;
            if (screen) {
                float* thr = m->thr + (size_t)base * m->max_q;                    // per (pair, split, query)
                int* ccnt = m->cand_cnt + (size_t)base * m->max_q;
                int2* cand = m->cand + (size_t)base * m->max_q * kCandCap;
                static const bool screen_v1 = getenv("MIF_MATCH_SCREEN_V1") != nullptr;      // A/B switch
                static const char* wide_env = getenv("MIF_MATCH_SCREEN_WIDE");               // A/B switch: 0 / 1
                // 512 queries per work item where a pair has enough queries to fill such items and the
                // database is large enough for its traffic to matter
                const bool wide = wide_env ? atoi(wide_env) != 0 : false;
                static const char* turns_env = getenv("MIF_MATCH_SCREEN_TURNS");             // A/B switch: 0 / 1
                const bool large = !device_counts && nq >= kTurnsMinQueries && nd >= kTurnsMinRows &&
                                   (double)nq * nd >= kTurnsMinScores;
                const bool turns = m->dbf != nullptr && (turns_env ? atoi(turns_env) != 0 : large);
                __bf16* dbf = m->dbf ? m->dbf + (size_t)base * m->rows_pad * (128 + 24) : nullptr;
                // Pass 0 of the second and third form looks at every n-th chunk of a split only. The limit it
                // leaves is looser, so a query's lists hold more rows (about 2 n more): n follows the
                // number of lists a query's candidates spread over (measured: 16 splits take n = 4, 8 take
                // 2, 4 take none; beyond 8 the lists overflow more often than the pass gets cheaper).
                static const char* sample_env = getenv("MIF_MATCH_SAMPLE");                  // tuning override
                const int per_split = ((nd + kDC - 1) / kDC + nsplit - 1) / nsplit;
                // Large searches only: with one list per query (frame batches) any sampling overflows lists
                // (57.3 -> 57.7 ms per step at n = 2), and a few thousand queries do not repay the longer lists
                // (2048 x 1 M rows: 2.5 -> 3.7 ms).
                const int sample = sample_env ? std::max(1, atoi(sample_env))
                                              : (per_split >= 64 && large ? std::max(1, std::min(nsplit / 4, 8)) : 1);
                m->last_passes = 1.0 + 1.0 / (screen_v1 ? 1 : sample);
#define MIF_LAUNCH_SCREEN(KK, V, PASS)                                                                           \
    do {                                                                                                         \
        if (screen_v1)                                                                                           \
            hipLaunchKernelGGL((k_match_screen<KK, V, PASS>), dim3(m->blocks), dim3(256), 0, st, tbl, np, nq, nd, \
                               dim, ld, (const float*)qn, (const float*)dn, (const unsigned int*)nmax, nsplit,   \
                               thr, ccnt, cand, m->work + 1 + PASS);                                             \
        else if (turns) {                                                                                        \
            if (PASS == 0)                                                                                       \
                hipLaunchKernelGGL((k_match_pack_rows<KK, V>),                                                   \
                                   dim3((unsigned)(((size_t)m->rows_pad * ((KK + 24) / 8) + 255) / 256), np),   \
                                   dim3(256), 0, st, tbl, nd, m->rows_pad, dim, ld, (const float*)dn, dbf);      \
            hipLaunchKernelGGL((k_match_screen4<KK, V, PASS>), dim3(m->blocks / 2), dim3(512), 0, st, tbl, np,  \
                               nq, nd, dim, ld, (const float*)qn, (const __bf16*)dbf, m->rows_pad,              \
                               (const unsigned int*)nmax, nsplit, thr, m->screen_slots, ccnt, cand,             \
                               m->work + 1 + PASS, sample);                                                      \
        }                                                                                                        \
        else if (wide)                                                                                           \
            hipLaunchKernelGGL((k_match_screen2<KK, V, PASS, 8>), dim3(m->blocks / 2), dim3(512), 0, st, tbl, np,  \
                               nq, nd, dim, ld, (const float*)qn, (const float*)dn, (const unsigned int*)nmax,   \
                               nsplit, thr, m->screen_slots, ccnt, cand, m->work + 1 + PASS,                    \
                               sample);                                                                          \
        else                                                                                                     \
            hipLaunchKernelGGL((k_match_screen2<KK, V, PASS, 4>), dim3(m->blocks), dim3(256), 0, st, tbl, np,    \
                               nq, nd, dim, ld, (const float*)qn, (const float*)dn, (const unsigned int*)nmax,   \
                               nsplit, thr, m->screen_slots, ccnt, cand, m->work + 1 + PASS,                    \
                               sample);                                                                          \
    } while (0)
#define MIF_LAUNCH_LIMITS()                                                                                      \
    do {                                                                                                         \
        if (!screen_v1)                                                                                          \
            hipLaunchKernelGGL(k_match_limits, dim3((nq + 255) / 256, np), dim3(256), 0, st, tbl, nq, nd,        \
                               (const float*)qn, (const unsigned int*)nmax, nsplit, thr, m->screen_slots);       \
    } while (0)
                if (dim <= 64) {
                    if (vec) { MIF_LAUNCH_SCREEN(64, true, 0); MIF_LAUNCH_LIMITS(); MIF_LAUNCH_SCREEN(64, true, 1); }
                    else     { MIF_LAUNCH_SCREEN(64, false, 0); MIF_LAUNCH_LIMITS(); MIF_LAUNCH_SCREEN(64, false, 1); }
                } else {
                    if (vec) { MIF_LAUNCH_SCREEN(128, true, 0); MIF_LAUNCH_LIMITS(); MIF_LAUNCH_SCREEN(128, true, 1); }
                    else     { MIF_LAUNCH_SCREEN(128, false, 0); MIF_LAUNCH_LIMITS(); MIF_LAUNCH_SCREEN(128, false, 1); }
                }
#undef MIF_LAUNCH_SCREEN
#undef MIF_LAUNCH_LIMITS
                // (work[3]: the number of queued queries; zeroed with the other counters above)
                // after a sampled first pass the lists are long: k_match_rank tightens the threshold first
                static const bool no_refilter = getenv("MIF_MATCH_NO_REFILTER") != nullptr;  // A/B switch
                const int refilter = (sample > 1 && !screen_v1 && !no_refilter) ? 1 : 0;
#define MIF_LAUNCH_RANK(V)                                                                                       \
    do {                                                                                                         \
        hipLaunchKernelGGL(k_match_rank<V>, dim3((nq + 127) / 128, np), dim3(128), 0, st, tbl, nq, nd, dim, ld,  \
                           (const float*)qn, (const float*)dn, nsplit, (const int*)ccnt, (const int2*)cand, ps, \
                           pi, m->ovf + (size_t)base * m->max_q, m->work + 3, (const unsigned int*)nmax,        \
                           refilter);                                                                            \
        hipLaunchKernelGGL(k_match_overflow<V>, dim3(2 * m->blocks), dim3(256), 0, st, tbl, nq, nd, dim, ld,     \
                           (const float*)qn, (const float*)dn, nsplit,                                          \
                           (const OverflowEntry*)(m->ovf + (size_t)base * m->max_q),                            \
                           (const unsigned int*)(m->work + 3), ps, pi);                                         \
    } while (0)
                if (vec) MIF_LAUNCH_RANK(true); else MIF_LAUNCH_RANK(false);
#undef MIF_LAUNCH_RANK
            } else {
#define MIF_LAUNCH_PARTIAL(KS, V)                                                                              \
    hipLaunchKernelGGL((k_match_partial<KS, V>), dim3(m->blocks), dim3(256), 0, st, tbl, np, nq, nd, dim, ld,    \
                       (const float*)qn, (const float*)dn, nsplit, ps, pi, m->work)
            if (dim <= 64) { if (vec) MIF_LAUNCH_PARTIAL(32, true); else MIF_LAUNCH_PARTIAL(32, false); }
            else           { if (vec) MIF_LAUNCH_PARTIAL(64, true); else MIF_LAUNCH_PARTIAL(64, false); }
#undef MIF_LAUNCH_PARTIAL
            }
        }
        if (vec_db && vec_q)
            hipLaunchKernelGGL(k_match_finalize<true>, dim3((nq + 127) / 128, np), dim3(128), 0, st, tbl, nq, nd, dim,
                               ld, screen ? -2 : nsplit, (const float*)ps, (const int*)pi);
        else
            hipLaunchKernelGGL(k_match_finalize<false>, dim3((nq + 127) / 128, np), dim3(128), 0, st, tbl, nq, nd, dim,
                               ld, screen ? -2 : nsplit, (const float*)ps, (const int*)pi);
        if (filter_mode != 0)
            hipLaunchKernelGGL(k_match_filter, dim3(np), dim3(256), 0, st, tbl, nq, filter_mode);
    }
    MIF_HIP(hipGetLastError());
    MIF_HIP(hipEventRecord(m->slot_ev[slot], st));
    return MIF_OK;
}

// ------------------------------------------------------- search without allocations
//
// mif_match_top2_device_noalloc: the exact search (k_row_norms, k_match_partial, k_match_finalize)
// for callers that cannot allocate. Everything a matcher handle would own lives in one fixed array in
// device memory: the pair table the kernels read, the work-item counters, the row norms and the
// per-split partial winners. Every GPU holds its own copy of the array at its own address; the
// address is asked for once per device (first call there) and kept.
constexpr int kNaMaxQuery = MIF_NOALLOC_MAX_QUERY;
constexpr int kNaMaxDatabase = MIF_NOALLOC_MAX_DATABASE;
constexpr size_t kNaOffTable = 0;                                                  // one mif_match_pair
constexpr size_t kNaOffWork = 256;                                                 // 4 counters
constexpr size_t kNaOffQn = 512;                                                   // float[max query]
constexpr size_t kNaOffDn = kNaOffQn + sizeof(float) * kNaMaxQuery;                // float[max database]
constexpr size_t kNaOffPartS = kNaOffDn + sizeof(float) * kNaMaxDatabase;          // float[splits][max query][2]
constexpr size_t kNaOffPartI = kNaOffPartS + sizeof(float) * 2 * kMaxSplit * kNaMaxQuery;
constexpr size_t kNaBytes = kNaOffPartI + sizeof(int) * 2 * kMaxSplit * kNaMaxQuery;
static_assert(sizeof(mif_match_pair) <= 256, "the pair table slot is 256 bytes");
static_assert(kNaOffDn % 256 == 0 && kNaOffPartS % 256 == 0 && kNaOffPartI % 256 == 0, "sub-buffers stay aligned");
static_assert(kNaBytes <= (size_t)8 << 20, "the fixed array is meant to stay small");

static __device__ __align__(256) unsigned char scratch[kNaBytes];

constexpr int kNaDevices = 64;
unsigned char* g_na_ptr[kNaDevices] = {};       // device address of `scratch` on every GPU seen so far
int g_na_blocks[kNaDevices] = {};               // grid of the search kernel there: two blocks per compute unit
std::mutex g_na_mutex;

// Device address of this GPU's copy of the array and the search kernel's grid; `device` is current.
// Both are asked for on the first call on a device and kept, so that later calls query nothing.
int noalloc_scratch(int device, unsigned char** out, int* blocks)
{
    std::lock_guard<std::mutex> lock(g_na_mutex);
    if (!g_na_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            mif::set_error("mif_match_top2_device_noalloc: hipGetSymbolAddress failed on device %d: %s", device,
                           hipGetErrorString(e));
            return MIF_ERR_HIP;
        }
        int cus = 0;
        g_na_blocks[device] = 512;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
            g_na_blocks[device] = 2 * cus;
        g_na_ptr[device] = static_cast<unsigned char*>(p);
    }
    *out = g_na_ptr[device];
    *blocks = g_na_blocks[device];
    return MIF_OK;
}

// The one-entry pair table is written by a kernel (the pair travels as a kernel argument), so that no
// host buffer has to be staged.
__global__ void k_write_pair(mif_match_pair* tbl, mif_match_pair pr) { *tbl = pr; }

} // namespace

extern "C" {

int mif_match_top2_device_noalloc(int device, const float* d_q, int nq, const float* d_db, int nd, int dim, int ld,
                                  int32_t* d_idx2, float* d_dist2, void* stream)
{
    if (device < 0 || device >= kNaDevices) {
        mif::set_error("mif_match_top2_device_noalloc: device %d out of range", device);
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if (!d_idx2 || !d_dist2 || (nq > 0 && !d_q) || (nd > 0 && !d_db)) {
        mif::set_error("mif_match_top2_device_noalloc: NULL argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if (nq < 0 || nd < 0 || nq > kNaMaxQuery || nd > kNaMaxDatabase) {
        mif::set_error("mif_match_top2_device_noalloc: %d x %d rows exceed the fixed workspace of %d x %d rows "
                       "(use a mif_matcher for larger searches)", nq, nd, kNaMaxQuery, kNaMaxDatabase);
        return MIF_ERR_CAPACITY;
    }
    if (dim < 1 || dim > 128 || ld < dim) {
        mif::set_error("match: dim must be 1..128 and ld >= dim (got dim %d, ld %d)", dim, ld);
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if (nq == 0) return MIF_OK;
    mif::DeviceGuard guard(device);
    if (!guard.ok) { mif::set_error("mif_match_top2_device_noalloc: cannot select device %d", device); return MIF_ERR_HIP; }
    unsigned char* base = nullptr;
    int blocks = 512;
    const int rc = noalloc_scratch(device, &base, &blocks);
    if (rc != MIF_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    mif_match_pair* tbl = reinterpret_cast<mif_match_pair*>(base + kNaOffTable);
    unsigned int* work = reinterpret_cast<unsigned int*>(base + kNaOffWork);
    float* qn = reinterpret_cast<float*>(base + kNaOffQn);
    float* dn = reinterpret_cast<float*>(base + kNaOffDn);
    float* ps = reinterpret_cast<float*>(base + kNaOffPartS);
    int* pi = reinterpret_cast<int*>(base + kNaOffPartI);
    mif_match_pair pr;
    memset(&pr, 0, sizeof(pr));
    pr.d_q = d_q; pr.d_db = d_db; pr.d_idx2 = d_idx2; pr.d_dist2 = d_dist2;
    hipLaunchKernelGGL(k_write_pair, dim3(1), dim3(1), 0, st, tbl, pr);
    const int nsplit = pick_split(nq, nd, 1, false);
    const bool vec_db = (ld % 4) == 0 && (reinterpret_cast<uintptr_t>(d_db) & 15u) == 0;
    const bool vec_q = (ld % 4) == 0 && (reinterpret_cast<uintptr_t>(d_q) & 15u) == 0;
    const bool vec = vec_db && vec_q;
    if (nd > 0) {
        // the counters of the last call are stale: they start from zero; norms and partial winners are
        // written before they are read for every row and split this call uses
        MIF_HIP(hipMemsetAsync(work, 0, 4 * sizeof(unsigned int), st));
        if (vec_db)
            hipLaunchKernelGGL(k_row_norms<true>, dim3((nd + 255) / 256, 1), dim3(256), 0, st, (PairTable)tbl, 1, nd, dim, ld, dn, (unsigned int*)nullptr);
        else
            hipLaunchKernelGGL(k_row_norms<false>, dim3((nd + 255) / 256, 1), dim3(256), 0, st, (PairTable)tbl, 1, nd, dim, ld, dn, (unsigned int*)nullptr);
        if (vec_q)
            hipLaunchKernelGGL(k_row_norms<true>, dim3((nq + 255) / 256, 1), dim3(256), 0, st, (PairTable)tbl, 0, nq, dim, ld, qn, (unsigned int*)nullptr);
        else
            hipLaunchKernelGGL(k_row_norms<false>, dim3((nq + 255) / 256, 1), dim3(256), 0, st, (PairTable)tbl, 0, nq, dim, ld, qn, (unsigned int*)nullptr);
#define MIF_LAUNCH_PARTIAL(KS, V)                                                                              \
    hipLaunchKernelGGL((k_match_partial<KS, V>), dim3(blocks), dim3(256), 0, st, (PairTable)tbl, 1, nq, nd, dim, ld, \
                       (const float*)qn, (const float*)dn, nsplit, ps, pi, work)
        if (dim <= 64) { if (vec) MIF_LAUNCH_PARTIAL(32, true); else MIF_LAUNCH_PARTIAL(32, false); }
        else           { if (vec) MIF_LAUNCH_PARTIAL(64, true); else MIF_LAUNCH_PARTIAL(64, false); }
#undef MIF_LAUNCH_PARTIAL
    }
    if (vec)
        hipLaunchKernelGGL(k_match_finalize<true>, dim3((nq + 127) / 128, 1), dim3(128), 0, st, (PairTable)tbl, nq, nd, dim,
                           ld, nsplit, (const float*)ps, (const int*)pi);
    else
        hipLaunchKernelGGL(k_match_finalize<false>, dim3((nq + 127) / 128, 1), dim3(128), 0, st, (PairTable)tbl, nq, nd, dim,
                           ld, nsplit, (const float*)ps, (const int*)pi);
    MIF_HIP(hipGetLastError());
    return MIF_OK;
}

int mif_match_create_batched(int device, int max_query, int max_database, int max_pairs, mif_matcher** out)
{
    if (!out) return MIF_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (max_query < 1 || max_database < 1 || max_pairs < 1) {
        mif::set_error("mif_match_create: capacities must be positive");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(device);
    if (!guard.ok) { mif::set_error("mif_match_create: cannot select device %d", device); return MIF_ERR_HIP; }
    mif_matcher* m = new mif_matcher();
    m->device = device; m->max_q = max_query; m->max_d = max_database; m->max_pairs = max_pairs;
    auto fail = [&](hipError_t e) {
        mif::set_error("mif_match_create: %s", hipGetErrorString(e));
        free_matcher(m);
        delete m;
        return MIF_ERR_HIP;
    };
    const size_t P = (size_t)max_pairs;
    hipError_t e;
    if ((e = hipMalloc(&m->qn, P * max_query * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&m->dn, P * max_database * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&m->part_s, P * kMaxSplit * max_query * 2 * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&m->part_i, P * kMaxSplit * max_query * 2 * sizeof(int))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&m->work, 4 * sizeof(unsigned int))) != hipSuccess) return fail(e);
    // room for a batch without database splits or for one pair with every split
    m->screen_slots = std::max<size_t>(P, kMaxSplit) * (size_t)max_query;
    if ((e = hipMalloc(&m->dnmax, P * sizeof(unsigned int))) != hipSuccess) return fail(e);
    // two values per (pair, split, query): the bound on the second smallest score, then the smallest score
    if ((e = hipMalloc(&m->thr, 2 * m->screen_slots * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&m->cand_cnt, m->screen_slots * sizeof(int))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&m->cand, m->screen_slots * kCandCap * sizeof(int2))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&m->ovf, P * max_query * sizeof(OverflowEntry))) != hipSuccess) return fail(e);
    // the bf16 image of the database rows for the screening passes of large searches (kTurnsMinScores);
    // a matcher too small for those has none, and without it (out of memory) the passes stage the fp32
    // rows themselves
    m->rows_pad = (max_database + kDC - 1) / kDC * kDC;
    const bool large = max_query >= kTurnsMinQueries && max_database >= kTurnsMinRows &&
                       (double)max_query * max_database >= kTurnsMinScores;
    if ((large || getenv("MIF_MATCH_SCREEN_TURNS")) &&
        hipMalloc(&m->dbf, P * (size_t)m->rows_pad * (128 + 24) * sizeof(__bf16)) != hipSuccess) {
        (void)hipGetLastError();
        m->dbf = nullptr;
    }
    for (int i = 0; i < kTableSlots; i++) {
        if ((e = hipHostMalloc(&m->h_tbl[i], P * sizeof(mif_match_pair))) != hipSuccess) return fail(e);
        if ((e = hipMalloc(&m->d_tbl[i], P * sizeof(mif_match_pair))) != hipSuccess) return fail(e);
        if ((e = hipEventCreateWithFlags(&m->slot_ev[i], hipEventDisableTiming)) != hipSuccess) return fail(e);
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
        m->blocks = 2 * cus;
    if (const char* e = getenv("MIF_MATCH_BLOCKS")) {        // tuning override: persistent blocks in total
        const int n = atoi(e);
        if (n >= 1 && n <= 65535) m->blocks = n;
    }
    *out = m;
    return MIF_OK;
}

int mif_match_create(int device, int max_query, int max_database, mif_matcher** out)
{
    return mif_match_create_batched(device, max_query, max_database, 1, out);
}

int mif_match_destroy(mif_matcher* m)
{
    if (!m) return MIF_OK;
    mif::DeviceGuard guard(m->device);
    (void)hipDeviceSynchronize();
    free_matcher(m);
    delete m;
    return MIF_OK;
}

int mif_match_set_screening(mif_matcher* m, int enable)
{
    if (!m) { mif::set_error("mif_match_set_screening: NULL handle"); return MIF_ERR_INVALID_ARGUMENT; }
    m->screening = enable != 0;
    return MIF_OK;
}

int mif_match_last_screen_passes(mif_matcher* m, double* passes)
{
    if (!m || !passes) { mif::set_error("mif_match_last_screen_passes: NULL argument"); return MIF_ERR_INVALID_ARGUMENT; }
    *passes = m->last_passes;
    return MIF_OK;
}

int mif_match_top2_batch_device(mif_matcher* m, const mif_match_pair* pairs, int npairs, int nq, int nd, int dim,
                                int ld, int filter_mode, void* stream)
{
    if (!m || (npairs > 0 && !pairs)) {
        mif::set_error("mif_match_top2_batch_device: NULL argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif::DeviceGuard guard(m->device);
    return match_batch(m, pairs, npairs, nq, nd, dim, ld, filter_mode, (hipStream_t)stream);
}

int mif_match_top2_device(mif_matcher* m, const float* d_q, int nq, const int32_t* d_nq, const float* d_db, int nd,
                          const int32_t* d_nd, int dim, int ld, int32_t* d_idx2, float* d_dist2, void* stream)
{
    if (!m || !d_idx2 || !d_dist2 || (nq > 0 && !d_q) || (nd > 0 && !d_db)) {
        mif::set_error("mif_match_top2_device: NULL argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    mif_match_pair pr;
    memset(&pr, 0, sizeof(pr));
    pr.d_q = d_q; pr.d_db = d_db; pr.d_nq = d_nq; pr.d_nd = d_nd; pr.d_idx2 = d_idx2; pr.d_dist2 = d_dist2;
    mif::DeviceGuard guard(m->device);
    return match_batch(m, &pr, 1, nq, nd, dim, ld, 0, (hipStream_t)stream);
}

int mif_match_top2(mif_matcher* m, const float* q, int nq, const float* db, int nd, int dim, int ld, int32_t* idx2,
                   float* dist2)
{
    return mif_match_top2_scores(m, q, nq, db, nd, dim, ld, idx2, dist2, nullptr);
}

int mif_match_top2_scores(mif_matcher* m, const float* q, int nq, const float* db, int nd, int dim, int ld,
                          int32_t* idx2, float* dist2, float* score2)
{
    if (!m || !idx2 || !dist2 || (nq > 0 && !q) || (nd > 0 && !db)) {
        mif::set_error("mif_match_top2: NULL argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if (nq < 0 || nd < 0 || nq > m->max_q || nd > m->max_d) {
        mif::set_error("mif_match_top2: %d x %d rows exceed the matcher's capacity %d x %d", nq, nd, m->max_q,
                       m->max_d);
        return MIF_ERR_CAPACITY;
    }
    if (nq == 0) return MIF_OK;
    mif::DeviceGuard guard(m->device);
    const size_t qb = (size_t)nq * ld * sizeof(float), db_b = (size_t)std::max(nd, 1) * ld * sizeof(float);
    if (qb > m->d_q_bytes) {
        (void)hipFree(m->d_q); m->d_q = nullptr; m->d_q_bytes = 0;
        MIF_HIP(hipMalloc(&m->d_q, qb));
        m->d_q_bytes = qb;
    }
    if (db_b > m->d_db_bytes) {
        (void)hipFree(m->d_db); m->d_db = nullptr; m->d_db_bytes = 0;
        MIF_HIP(hipMalloc(&m->d_db, db_b));
        m->d_db_bytes = db_b;
    }
    if (!m->d_idx) {
        MIF_HIP(hipMalloc(&m->d_idx, (size_t)m->max_q * 2 * sizeof(int)));
        MIF_HIP(hipMalloc(&m->d_dist, (size_t)m->max_q * 2 * sizeof(float)));
        MIF_HIP(hipMalloc(&m->d_score, (size_t)m->max_q * 2 * sizeof(float)));
    }
    MIF_HIP(hipMemcpy(m->d_q, q, qb, hipMemcpyHostToDevice));
    if (nd > 0) MIF_HIP(hipMemcpy(m->d_db, db, (size_t)nd * ld * sizeof(float), hipMemcpyHostToDevice));
    mif_match_pair pr;
    memset(&pr, 0, sizeof(pr));
    pr.d_q = m->d_q; pr.d_db = m->d_db; pr.d_idx2 = m->d_idx; pr.d_dist2 = m->d_dist;
    pr.d_score2 = score2 ? m->d_score : nullptr;
    const int rc = match_batch(m, &pr, 1, nq, nd, dim, ld, 0, nullptr);
    if (rc != MIF_OK) return rc;
    MIF_HIP(hipDeviceSynchronize());
    MIF_HIP(hipMemcpy(idx2, m->d_idx, (size_t)nq * 2 * sizeof(int), hipMemcpyDeviceToHost));
    MIF_HIP(hipMemcpy(dist2, m->d_dist, (size_t)nq * 2 * sizeof(float), hipMemcpyDeviceToHost));
    if (score2) MIF_HIP(hipMemcpy(score2, m->d_score, (size_t)nq * 2 * sizeof(float), hipMemcpyDeviceToHost));
    return MIF_OK;
}

} // extern "C"
