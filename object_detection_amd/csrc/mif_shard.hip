// Descriptor database sharded row-wise over several GPUs: every GPU searches its rows for the two nearest
// neighbours of every query (the matcher of mif_match.hip), the per-shard candidates are exchanged with one
// RCCL all-gather over xGMI and every GPU merges them into the global two nearest.
//
// The reference has no counterpart: its only split is the CPU/GPU NDRange split of clDriver.h:1879-1921,
// and its matcher is the FLANN call of common/kpmatch.h:174-176, 284-286. The contract is the exact
// search's: the merged result equals one search over the whole database, bit for bit, because a shard
// reports for each candidate the ranking score it won with, and the merge orders all candidates by
// (score, global row index) exactly as one search orders its database splits.
//
// A handle drives `nlocal` shards from one process (one device, matcher, stream and communicator rank
// each). Processes with one GPU each use nlocal = 1; one process that drives all GPUs of a node uses
// nlocal = nranks, and the collective calls of its shards are fused in one RCCL group. Shards that share a
// device (RCCL refuses two ranks on one GPU; only possible when one process holds every rank) exchange
// their candidates with device-to-device copies instead; that form exists so that the pack and merge
// kernels and the index arithmetic can run on a machine with one GPU.
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <vector>
#include <rccl/rccl.h>
#include "mif_common.h"

namespace {

// RCCL is loaded when the first sharded matcher is made, not when this library is loaded: callers that
// never shard a database do not pay for it, and a process that already holds a copy (PyTorch ships one
// under the same soname) gets that copy.
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

const Rccl* rccl()
{
    static Rccl r;
    static std::once_flag once;
    static bool ok = false;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
#define MIF_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, sym))
        MIF_SYM(GetUniqueId, "ncclGetUniqueId");
        MIF_SYM(CommInitRank, "ncclCommInitRank");
        MIF_SYM(CommDestroy, "ncclCommDestroy");
        MIF_SYM(AllGather, "ncclAllGather");
        MIF_SYM(GroupStart, "ncclGroupStart");
        MIF_SYM(GroupEnd, "ncclGroupEnd");
        MIF_SYM(GetErrorString, "ncclGetErrorString");
#undef MIF_SYM
        ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd &&
             r.GetErrorString;
    });
    if (!ok) {
        mif::set_error("RCCL (librccl.so.1) could not be loaded: %s", r.lib ? "symbols missing" : dlerror());
        return nullptr;
    }
    return &r;
}

#define MIF_NCCL(R, call)                                                                         \
    do {                                                                                          \
        ncclResult_t e_ = (call);                                                                 \
        if (e_ != ncclSuccess) {                                                                  \
            ::mif::set_error("%s failed: %s (%s:%d)", #call, (R)->GetErrorString(e_), __FILE__, __LINE__); \
            return MIF_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

constexpr int kMsgFloats = 6;   // per query: (score, global index bits, distance) of the two local winners
constexpr float kInf = __builtin_inff();

// Message of this shard: for each query the two local winners as (ranking score, GLOBAL row index,
// distance); a missing winner travels as (+inf, -1, +inf).
__global__ __launch_bounds__(256) void k_shard_pack(const int32_t* __restrict__ idx2, const float* __restrict__ dist2,
                                                    const float* __restrict__ score2, int offset, int nq,
                                                    float* __restrict__ msg)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int j = idx2[2 * i + k];
        const int g = j >= 0 ? j + offset : -1;
        msg[kMsgFloats * i + 3 * k + 0] = j >= 0 ? score2[2 * i + k] : kInf;
        msg[kMsgFloats * i + 3 * k + 1] = __builtin_bit_cast(float, g);
        msg[kMsgFloats * i + 3 * k + 2] = j >= 0 ? dist2[2 * i + k] : kInf;
    }
}

__device__ __forceinline__ bool before(float sa, int ia, float sb, int ib)
{
    return sa < sb || (sa == sb && ia < ib);
}

// all[rank][query][6] -> the two smallest (score, index) over all ranks, reported nearest first by
// (distance, index): what k_match_finalize does with the splits of one database.
__global__ __launch_bounds__(256) void k_shard_merge(const float* __restrict__ all, int nranks, int nq,
                                                     int32_t* __restrict__ idx2, float* __restrict__ dist2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    float s1 = kInf, s2 = kInf, e1 = kInf, e2 = kInf;
    int i1 = 0x7fffffff, i2 = 0x7fffffff;
    for (int r = 0; r < nranks; r++) {
        const float* m = all + ((size_t)r * nq + i) * kMsgFloats;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const float s = m[3 * k];
            const int j = __builtin_bit_cast(int, m[3 * k + 1]);
            const float e = m[3 * k + 2];
            if (j < 0) continue;
            if (before(s, j, s1, i1)) {
                s2 = s1; i2 = i1; e2 = e1;
                s1 = s; i1 = j; e1 = e;
            } else if (before(s, j, s2, i2)) {
                s2 = s; i2 = j; e2 = e;
            }
        }
    }
    int j1 = i1 != 0x7fffffff ? i1 : -1, j2 = i2 != 0x7fffffff ? i2 : -1;
    if (j1 < 0) e1 = kInf;
    if (j2 < 0) e2 = kInf;
    if (j2 >= 0 && (e2 < e1 || (e2 == e1 && j2 < j1))) {
        const float te = e1; e1 = e2; e2 = te;
        const int tj = j1; j1 = j2; j2 = tj;
    }
    idx2[2 * i] = j1; idx2[2 * i + 1] = j2;
    dist2[2 * i] = e1; dist2[2 * i + 1] = e2;
}

struct Shard {
    int device = 0, rank = 0;
    mif_matcher* matcher = nullptr;
    hipStream_t stream = nullptr;       // used when the caller passes none
    ncclComm_t comm = nullptr;
    int32_t* lidx = nullptr;            // [max_q][2] local winners
    float* ldist = nullptr;
    float* lscore = nullptr;
    float* send = nullptr;              // [max_q][6]
    float* recv = nullptr;              // [nranks][max_q][6]
    hipEvent_t packed = nullptr;        // device-copy exchange: the message is complete
    hipEvent_t taken = nullptr;         // device-copy exchange: this shard has copied every message
    hipEvent_t done = nullptr;          // end of the shard's part of the last search, on whichever stream it ran
};

} // namespace

struct mif_match_sharded {
    int nlocal = 0, nranks = 0, first_rank = 0, max_q = 0, max_rows = 0;
    bool device_copies = false;         // shards share a device: exchange by copies, no communicator
    bool searched = false;
    std::vector<Shard> s;
};

namespace {

void free_sharded(mif_match_sharded* h)
{
    const Rccl* R = h->device_copies ? nullptr : rccl();
    for (Shard& sh : h->s) {
        mif::DeviceGuard guard(sh.device);
        // the last search may have run on a stream of the caller's (mif_shard_io.stream): its end is an
        // event, which stays valid whatever happened to that stream since
        if (sh.done && h->searched) (void)hipEventSynchronize(sh.done);
        if (sh.stream) (void)hipStreamSynchronize(sh.stream);
        if (sh.comm && R) (void)R->CommDestroy(sh.comm);
        if (sh.matcher) (void)mif_match_destroy(sh.matcher);
        (void)hipFree(sh.lidx); (void)hipFree(sh.ldist); (void)hipFree(sh.lscore);
        (void)hipFree(sh.send); (void)hipFree(sh.recv);
        if (sh.packed) (void)hipEventDestroy(sh.packed);
        if (sh.taken) (void)hipEventDestroy(sh.taken);
        if (sh.done) (void)hipEventDestroy(sh.done);
        if (sh.stream) (void)hipStreamDestroy(sh.stream);
    }
}

} // namespace

extern "C" {

int mif_match_sharded_unique_id(void* id)
{
    if (!id) { mif::set_error("mif_match_sharded_unique_id: NULL argument"); return MIF_ERR_INVALID_ARGUMENT; }
    static_assert(MIF_SHARD_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is RCCL's");
    const Rccl* R = rccl();
    if (!R) return MIF_ERR_UNSUPPORTED;
    ncclUniqueId u;
    MIF_NCCL(R, R->GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return MIF_OK;
}

int mif_match_sharded_create(const int* devices, int nlocal, int first_rank, int nranks, const void* id,
                             int max_query, int max_shard_rows, mif_match_sharded** out)
{
    if (!out) return MIF_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!devices || nlocal < 1 || nranks < nlocal || first_rank < 0 || first_rank + nlocal > nranks ||
        max_query < 1 || max_shard_rows < 1) {
        mif::set_error("mif_match_sharded_create: bad argument (nlocal %d, first_rank %d, nranks %d, capacities "
                       "%d x %d)", nlocal, first_rank, nranks, max_query, max_shard_rows);
        return MIF_ERR_INVALID_ARGUMENT;
    }
    bool shared_device = false;
    for (int a = 0; a < nlocal; a++)
        for (int b = a + 1; b < nlocal; b++) shared_device = shared_device || devices[a] == devices[b];
    if (shared_device && nlocal != nranks) {
        mif::set_error("mif_match_sharded_create: shards may share a device only when this process holds all "
                       "%d ranks (it holds %d)", nranks, nlocal);
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if (!shared_device && nlocal != nranks && !id) {
        mif::set_error("mif_match_sharded_create: ranks in other processes need the id made by "
                       "mif_match_sharded_unique_id on one of them");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    const Rccl* R = nullptr;
    if (!shared_device) {
        R = rccl();
        if (!R) return MIF_ERR_UNSUPPORTED;
    }
    mif_match_sharded* h = new mif_match_sharded();
    h->nlocal = nlocal; h->nranks = nranks; h->first_rank = first_rank;
    h->max_q = max_query; h->max_rows = max_shard_rows; h->device_copies = shared_device;
    h->s.resize(nlocal);
    auto fail = [&](int rc) { free_sharded(h); delete h; return rc; };
    const size_t Q = (size_t)max_query;
    for (int i = 0; i < nlocal; i++) {
        Shard& sh = h->s[i];
        sh.device = devices[i];
        sh.rank = first_rank + i;
        mif::DeviceGuard guard(sh.device);
        if (!guard.ok) { mif::set_error("mif_match_sharded_create: cannot select device %d", sh.device); return fail(MIF_ERR_HIP); }
        const int rc = mif_match_create(sh.device, max_query, max_shard_rows, &sh.matcher);
        if (rc != MIF_OK) return fail(rc);
        hipError_t e = hipStreamCreateWithFlags(&sh.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(&sh.lidx, Q * 2 * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc(&sh.ldist, Q * 2 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&sh.lscore, Q * 2 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&sh.send, Q * kMsgFloats * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&sh.recv, (size_t)nranks * Q * kMsgFloats * sizeof(float));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sh.packed, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sh.taken, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sh.done, hipEventDisableTiming);
        if (e != hipSuccess) {
            mif::set_error("mif_match_sharded_create: %s", hipGetErrorString(e));
            return fail(MIF_ERR_HIP);
        }
    }
    if (!shared_device) {
        ncclUniqueId u;
        if (id) {
            memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
        } else {
            const ncclResult_t e = R->GetUniqueId(&u);
            if (e != ncclSuccess) { mif::set_error("ncclGetUniqueId: %s", R->GetErrorString(e)); return fail(MIF_ERR_HIP); }
        }
        // the ranks of one process join together: one group around their init calls
        ncclResult_t e = R->GroupStart();
        for (int i = 0; i < nlocal && e == ncclSuccess; i++) {
            Shard& sh = h->s[i];
            mif::DeviceGuard guard(sh.device);
            e = R->CommInitRank(&sh.comm, nranks, u, sh.rank);
        }
        const ncclResult_t e2 = R->GroupEnd();
        if (e == ncclSuccess) e = e2;
        if (e != ncclSuccess) {
            mif::set_error("mif_match_sharded_create: ncclCommInitRank: %s", R->GetErrorString(e));
            return fail(MIF_ERR_HIP);
        }
    }
    *out = h;
    return MIF_OK;
}

int mif_match_sharded_destroy(mif_match_sharded* h)
{
    if (!h) return MIF_OK;
    free_sharded(h);
    delete h;
    return MIF_OK;
}

int mif_match_sharded_ranks(const mif_match_sharded* h, int* nlocal, int* nranks, int* uses_rccl)
{
    if (!h) return MIF_ERR_INVALID_ARGUMENT;
    if (nlocal) *nlocal = h->nlocal;
    if (nranks) *nranks = h->nranks;
    if (uses_rccl) *uses_rccl = h->device_copies ? 0 : 1;
    return MIF_OK;
}

int mif_match_sharded_set_screening(mif_match_sharded* h, int enable)
{
    if (!h) { mif::set_error("mif_match_sharded_set_screening: NULL handle"); return MIF_ERR_INVALID_ARGUMENT; }
    for (Shard& sh : h->s) {
        const int rc = mif_match_set_screening(sh.matcher, enable);
        if (rc != MIF_OK) return rc;
    }
    return MIF_OK;
}

int mif_match_sharded_last_screen_passes(mif_match_sharded* h, double* passes)
{
    if (!h || !passes || h->s.empty()) {
        mif::set_error("mif_match_sharded_last_screen_passes: bad argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    return mif_match_last_screen_passes(h->s[0].matcher, passes);
}

int mif_match_sharded_search(mif_match_sharded* h, const mif_shard_io* io, int nq, int dim, int ld)
{
    if (!h || !io) { mif::set_error("mif_match_sharded_search: NULL argument"); return MIF_ERR_INVALID_ARGUMENT; }
    if (nq < 0 || nq > h->max_q) {
        mif::set_error("mif_match_sharded_search: %d queries exceed the capacity of %d", nq, h->max_q);
        return MIF_ERR_CAPACITY;
    }
    for (int i = 0; i < h->nlocal; i++) {
        if (io[i].shard_rows < 0 || io[i].shard_rows > h->max_rows) {
            mif::set_error("mif_match_sharded_search: shard %d has %d rows, capacity %d", i, io[i].shard_rows, h->max_rows);
            return MIF_ERR_CAPACITY;
        }
        if (!io[i].d_idx2 || !io[i].d_dist2 || (nq > 0 && !io[i].d_q) || (io[i].shard_rows > 0 && !io[i].d_shard) ||
            io[i].global_offset < 0) {
            mif::set_error("mif_match_sharded_search: bad argument for shard %d", i);
            return MIF_ERR_INVALID_ARGUMENT;
        }
    }
    // everything the local searches would refuse is refused here, before the first shard enqueues:
    // a rank that returned between its enqueue and the collective would leave the other ranks waiting
    if (dim < 1 || dim > 128 || ld < dim) {
        mif::set_error("mif_match_sharded_search: dim must be 1..128 and ld >= dim (got dim %d, ld %d)", dim, ld);
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if (nq == 0) return MIF_OK;
    const Rccl* R = nullptr;
    if (!h->device_copies) {
        R = rccl();
        if (!R) return MIF_ERR_UNSUPPORTED;
    }
    const size_t msg_floats = (size_t)nq * kMsgFloats;
    // 1. local search and message of every shard
    for (int i = 0; i < h->nlocal; i++) {
        Shard& sh = h->s[i];
        mif::DeviceGuard guard(sh.device);
        hipStream_t st = io[i].stream ? (hipStream_t)io[i].stream : sh.stream;
        if (h->device_copies && h->searched) {
            // the message buffer is rewritten below: every shard must have copied the last one
            for (int j = 0; j < h->nlocal; j++) MIF_HIP(hipStreamWaitEvent(st, h->s[j].taken, 0));
        }
        mif_match_pair pr;
        memset(&pr, 0, sizeof(pr));
        pr.d_q = io[i].d_q; pr.d_db = io[i].d_shard;
        pr.d_idx2 = sh.lidx; pr.d_dist2 = sh.ldist; pr.d_score2 = sh.lscore;
        const int rc = mif_match_top2_batch_device(sh.matcher, &pr, 1, nq, io[i].shard_rows, dim, ld, MIF_FILTER_NONE, st);
        if (rc != MIF_OK) return rc;
        hipLaunchKernelGGL(k_shard_pack, dim3((nq + 255) / 256), dim3(256), 0, st, (const int32_t*)sh.lidx,
                           (const float*)sh.ldist, (const float*)sh.lscore, io[i].global_offset, nq, sh.send);
        MIF_HIP(hipGetLastError());
        if (h->device_copies) MIF_HIP(hipEventRecord(sh.packed, st));
    }
    // 2. exchange: all-gather of the messages (rank-major in every shard's receive buffer)
    if (h->device_copies) {
        for (int i = 0; i < h->nlocal; i++) {
            Shard& sh = h->s[i];
            mif::DeviceGuard guard(sh.device);
            hipStream_t st = io[i].stream ? (hipStream_t)io[i].stream : sh.stream;
            for (int j = 0; j < h->nlocal; j++) {
                MIF_HIP(hipStreamWaitEvent(st, h->s[j].packed, 0));
                MIF_HIP(hipMemcpyAsync(sh.recv + (size_t)h->s[j].rank * msg_floats, h->s[j].send,
                                       msg_floats * sizeof(float), hipMemcpyDeviceToDevice, st));
            }
            MIF_HIP(hipEventRecord(sh.taken, st));
        }
    } else {
        MIF_NCCL(R, R->GroupStart());
        ncclResult_t e = ncclSuccess;
        for (int i = 0; i < h->nlocal && e == ncclSuccess; i++) {
            Shard& sh = h->s[i];
            mif::DeviceGuard guard(sh.device);
            hipStream_t st = io[i].stream ? (hipStream_t)io[i].stream : sh.stream;
            e = R->AllGather(sh.send, sh.recv, msg_floats, ncclFloat, sh.comm, st);
        }
        const ncclResult_t e2 = R->GroupEnd();
        if (e == ncclSuccess) e = e2;
        if (e != ncclSuccess) {
            mif::set_error("mif_match_sharded_search: ncclAllGather: %s", R->GetErrorString(e));
            return MIF_ERR_HIP;
        }
    }
    // 3. merge on every shard
    for (int i = 0; i < h->nlocal; i++) {
        Shard& sh = h->s[i];
        mif::DeviceGuard guard(sh.device);
        hipStream_t st = io[i].stream ? (hipStream_t)io[i].stream : sh.stream;
        hipLaunchKernelGGL(k_shard_merge, dim3((nq + 255) / 256), dim3(256), 0, st, (const float*)sh.recv, h->nranks, nq,
                           io[i].d_idx2, io[i].d_dist2);
        MIF_HIP(hipGetLastError());
        MIF_HIP(hipEventRecord(sh.done, st));
    }
    h->searched = true;
    return MIF_OK;
}

int mif_match_sharded_sync(mif_match_sharded* h)
{
    if (!h) return MIF_ERR_INVALID_ARGUMENT;
    for (Shard& sh : h->s) {
        mif::DeviceGuard guard(sh.device);
        if (h->searched) MIF_HIP(hipEventSynchronize(sh.done));     // the last search, wherever it ran
        MIF_HIP(hipStreamSynchronize(sh.stream));
    }
    return MIF_OK;
}

} // extern "C"
