// Frame-sequence pipeline over the batched SIFT handles and the batched matcher:
// the loop of SIFT/main.cpp:103-153 (-v: one extract_features call per frame,
// features of the previous frame kept for comparison) for whole batches of
// frames that stay on the GPU, with every frame matched against its predecessor
// under the good-match rule of common/kpmatch.h:177-202.
//
// Schedule of batch k (two SIFT handles used alternately and the matcher, all on
// ONE stream; uploads of host frames on a second one):
//
//     pyramid(k)  ->  matcher(k - 1)  ->  features(k)  ->  pyramid(k + 1)  -> ...
//     upload(k + 1) beside all of it, from the moment pyramid(k - 1) has read its frames
//
// Every kernel of the chain fills the GPU, so running the matcher beside the
// feature phase on a stream of its own gains nothing (measured), and it costs
// the uploads their overlap: a process has few hardware queues (4 by default)
// for all its streams, a stream that waits for an event blocks its whole queue,
// and a matcher stream waiting for pyramid(k) held up the copy stream's
// commands behind it (1080p BGR, batch 512: 100 ms per step against 63 ms with
// one compute stream). Frame 0 of a batch is matched against the last frame of
// the batch before, whose descriptors are copied aside when that batch finishes.
// Nothing waits on the host until mif_video_collect.
#include <cstdint>
#include <cstring>
#include <deque>
#include <vector>
#include "mif_common.h"

namespace {

constexpr int kCntWords = 8;            // counts row of a frame: n_oct0..2, n_total, raw_oct0..2, status
constexpr int kCntTotal = 3, kCntStatus = 7;
constexpr int kMatchDim = 127, kMatchLd = 128;   // the shifted descriptor row of kpmatch.h:15,127-128

// host copies of one submitted batch, filled by asynchronous copies on the compute stream
struct BatchRecord {
    int nframes = 0;
    bool first_has_previous = false;    // frame 0 was matched (a frame preceded it)
    bool matched = true;                // the batch went through the matcher
    int32_t* h_counts = nullptr;        // pinned [batch][8]
    int32_t* h_summary = nullptr;       // pinned [batch][2] = {kept, query rows}
};

} // namespace

struct mif_video {
    int device = 0, W = 0, H = 0, C = 0, batch = 0, cap = 0;
    mif_sift* sift[2] = {nullptr, nullptr};
    mif_matcher* matcher = nullptr;
    hipStream_t xstream = nullptr;      // everything but the uploads: both handles, matcher, result downloads
                                        // (the first handle's own stream; not owned here)
    hipStream_t cstream = nullptr;      // uploads of host frames
    // matcher outputs per handle parity
    int32_t* idx[2] = {nullptr, nullptr};       // [batch][cap][2]
    float* dist[2] = {nullptr, nullptr};        // [batch][cap][2]
    uint8_t* good[2] = {nullptr, nullptr};      // [batch][cap]
    int32_t* summ[2] = {nullptr, nullptr};      // [batch][2]
    // descriptors and count of the last frame of a finished batch; three in rotation: one is
    // written while the matcher may still read the one from two batches ago
    float* snap_des[3] = {nullptr, nullptr, nullptr};
    int32_t* snap_cnt[3] = {nullptr, nullptr, nullptr};
    uint8_t* d_frames[2] = {nullptr, nullptr};  // upload targets of mif_video_submit_host (allocated on first use)
    hipEvent_t upload_ev[2] = {nullptr, nullptr};   // end of the last upload into d_frames[i]
    bool upload_pending[2] = {false, false};
    long long step = 0;                 // batches submitted
    bool matching = true;               // frames are matched against their predecessors
    bool pending = false;               // the last submitted batch has not been matched yet
    int pending_frames = 0;
    std::deque<BatchRecord> inflight;   // submitted, not collected (oldest first)
    std::vector<BatchRecord> pool;      // records free for reuse
    std::vector<mif_match_pair> pairs;
};

namespace {

void free_record(BatchRecord& r)
{
    if (r.h_counts) (void)hipHostFree(r.h_counts);
    if (r.h_summary) (void)hipHostFree(r.h_summary);
    r.h_counts = nullptr;
    r.h_summary = nullptr;
}

void free_all(mif_video* v)
{
    for (auto& r : v->inflight) free_record(r);
    for (auto& r : v->pool) free_record(r);
    for (int i = 0; i < 2; i++) {
        (void)hipFree(v->idx[i]); (void)hipFree(v->dist[i]); (void)hipFree(v->good[i]); (void)hipFree(v->summ[i]);
        (void)hipFree(v->d_frames[i]);
        if (v->upload_ev[i]) (void)hipEventDestroy(v->upload_ev[i]);
        if (v->sift[i]) (void)mif_sift_destroy(v->sift[i]);
    }
    for (int i = 0; i < 3; i++) { (void)hipFree(v->snap_des[i]); (void)hipFree(v->snap_cnt[i]); }
    if (v->matcher) (void)mif_match_destroy(v->matcher);
    if (v->cstream) (void)hipStreamDestroy(v->cstream);
}

int get_record(mif_video* v, BatchRecord* out)
{
    if (!v->pool.empty()) {
        *out = v->pool.back();
        v->pool.pop_back();
        return MIF_OK;
    }
    BatchRecord r;
    MIF_HIP(hipHostMalloc(&r.h_counts, (size_t)v->batch * kCntWords * sizeof(int32_t)));
    if (hipHostMalloc(&r.h_summary, (size_t)v->batch * 2 * sizeof(int32_t)) != hipSuccess) {
        (void)hipHostFree(r.h_counts);
        mif::set_error("mif_video: out of pinned host memory");
        return MIF_ERR_HIP;
    }
    *out = r;
    return MIF_OK;
}

// Matches batch k (already extracted on handle k % 2) and downloads its
// counts and summaries into the batch's record (the newest in flight).
int enqueue_match(mif_video* v, long long k, int nf)
{
    const int hp = (int)(k % 2);
    void *d_des = nullptr, *d_cnt = nullptr;
    int rc = mif_sift_device_results(v->sift[hp], nullptr, nullptr, &d_des, &d_cnt);
    if (rc != MIF_OK) return rc;
    const float* des = (const float*)d_des;
    const int32_t* cnt = (const int32_t*)d_cnt;
    const size_t row = (size_t)v->cap * 128;
    BatchRecord& rec = v->inflight.back();
    v->pairs.clear();
    for (int b = 0; b < nf; b++) {
        mif_match_pair p;
        memset(&p, 0, sizeof(p));
        if (b == 0) {
            if (k == 0) continue;                            // the very first frame has no predecessor
            const int sn = (int)((k - 1) % 3);
            p.d_db = v->snap_des[sn];
            p.d_nd = v->snap_cnt[sn];
        } else {
            p.d_db = des + (size_t)(b - 1) * row;
            p.d_nd = cnt + (size_t)(b - 1) * kCntWords + kCntTotal;
        }
        p.d_q = des + (size_t)b * row;
        p.d_nq = cnt + (size_t)b * kCntWords + kCntTotal;
        p.d_idx2 = v->idx[hp] + (size_t)b * v->cap * 2;
        p.d_dist2 = v->dist[hp] + (size_t)b * v->cap * 2;
        p.d_good = v->good[hp] + (size_t)b * v->cap;
        p.d_summary = v->summ[hp] + (size_t)b * 2;
        v->pairs.push_back(p);
    }
    rec.first_has_previous = k > 0;
    if (!rec.matched) v->pairs.clear();          // the flag was taken when the batch was submitted
    if (!v->pairs.empty()) {
        rc = mif_match_top2_batch_device(v->matcher, v->pairs.data(), (int)v->pairs.size(), v->cap, v->cap, kMatchDim,
                                         kMatchLd, MIF_FILTER_SIFT, v->xstream);
        if (rc != MIF_OK) return rc;
    }
    MIF_HIP(hipMemcpyAsync(rec.h_counts, cnt, (size_t)nf * kCntWords * sizeof(int32_t), hipMemcpyDeviceToHost,
                           v->xstream));
    MIF_HIP(hipMemcpyAsync(rec.h_summary, v->summ[hp], (size_t)nf * 2 * sizeof(int32_t), hipMemcpyDeviceToHost,
                           v->xstream));
    return MIF_OK;
}

int submit(mif_video* v, const void* d_frames, int nframes)
{
    const long long k = v->step;
    mif_sift* h = v->sift[k % 2];
    // stream order does the rest: the batch before this one and the matcher that read this handle's
    // results two batches back are on the same stream
    int rc = mif_sift_enqueue_pyramid(h, d_frames, nframes);
    if (rc != MIF_OK) return rc;
    if (v->pending) {
        // (the record of batch k - 1 is the newest in flight: the one of batch k is added below)
        rc = enqueue_match(v, k - 1, v->pending_frames);
        if (rc != MIF_OK) return rc;
        v->pending = false;
    }
    rc = mif_sift_enqueue_features(h, 1);
    if (rc != MIF_OK) return rc;
    rc = mif_sift_snapshot_frame(h, nframes - 1, v->snap_des[k % 3], v->snap_cnt[k % 3]);
    if (rc != MIF_OK) return rc;
    BatchRecord rec;
    rc = get_record(v, &rec);
    if (rc != MIF_OK) return rc;
    rec.nframes = nframes;
    rec.matched = v->matching;
    v->inflight.push_back(rec);
    v->pending = true;
    v->pending_frames = nframes;
    v->step++;
    return MIF_OK;
}

} // namespace

extern "C" {

int mif_video_create(int device, int width, int height, int channels, int batch, mif_video** out)
{
    if (!out) { mif::set_error("mif_video_create: out is NULL"); return MIF_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    if (batch < 1) { mif::set_error("mif_video_create: batch %d", batch); return MIF_ERR_INVALID_ARGUMENT; }
    mif::DeviceGuard guard(device);
    if (!guard.ok) { mif::set_error("mif_video_create: cannot select device %d", device); return MIF_ERR_HIP; }
    mif_video* v = new mif_video();
    v->device = device; v->W = width; v->H = height; v->C = channels; v->batch = batch;
    auto fail = [&](int rc) { free_all(v); delete v; return rc; };
    for (int i = 0; i < 2; i++) {
        const int rc = mif_sift_create_batched(device, width, height, channels, 0, batch, &v->sift[i]);
        if (rc != MIF_OK) return fail(rc);
    }
    // one compute stream for both handles and the matcher (see the head of this file)
    v->xstream = (hipStream_t)mif_sift_stream(v->sift[0]);
    {
        const int rc = mif_sift_set_stream(v->sift[1], v->xstream);
        if (rc != MIF_OK) return fail(rc);
    }
    v->cap = mif_sift_capacity(v->sift[0]);
    const size_t cap = (size_t)(v->cap > 0 ? v->cap : 1), B = (size_t)batch;
    int rc = mif_match_create_batched(device, (int)cap, (int)cap, batch, &v->matcher);
    if (rc != MIF_OK) return fail(rc);
    hipError_t e = hipStreamCreateWithFlags(&v->cstream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipMalloc(&v->idx[i], B * cap * 2 * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc(&v->dist[i], B * cap * 2 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&v->good[i], B * cap);
        if (e == hipSuccess) e = hipMalloc(&v->summ[i], B * 2 * sizeof(int32_t));
        if (e == hipSuccess) e = hipMemset(v->summ[i], 0, B * 2 * sizeof(int32_t));
    }
    for (int i = 0; i < 3 && e == hipSuccess; i++) {
        e = hipMalloc(&v->snap_des[i], cap * 128 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&v->snap_cnt[i], sizeof(int32_t));
        if (e == hipSuccess) e = hipMemset(v->snap_cnt[i], 0, sizeof(int32_t));
    }
    if (e != hipSuccess) {
        mif::set_error("mif_video_create: %s", hipGetErrorString(e));
        return fail(MIF_ERR_HIP);
    }
    *out = v;
    return MIF_OK;
}

int mif_video_destroy(mif_video* v)
{
    if (!v) return MIF_OK;
    mif::DeviceGuard guard(v->device);
    for (int i = 0; i < 2; i++)
        if (v->sift[i]) (void)mif_sift_sync(v->sift[i]);
    if (v->cstream) (void)hipStreamSynchronize(v->cstream);
    free_all(v);
    delete v;
    return MIF_OK;
}

int mif_video_submit_device(mif_video* v, const void* d_frames, int nframes)
{
    if (!v || !d_frames) { mif::set_error("mif_video_submit_device: NULL argument"); return MIF_ERR_INVALID_ARGUMENT; }
    if (nframes < 1 || nframes > v->batch) {
        mif::set_error("mif_video_submit: %d frames do not fit the pipeline's batch of %d", nframes, v->batch);
        return MIF_ERR_CAPACITY;
    }
    mif::DeviceGuard guard(v->device);
    return submit(v, d_frames, nframes);
}

int mif_video_submit_host(mif_video* v, const uint8_t* frames, int nframes)
{
    if (!v || !frames) { mif::set_error("mif_video_submit_host: NULL argument"); return MIF_ERR_INVALID_ARGUMENT; }
    if (nframes < 1 || nframes > v->batch) {
        mif::set_error("mif_video_submit: %d frames do not fit the pipeline's batch of %d", nframes, v->batch);
        return MIF_ERR_CAPACITY;
    }
    mif::DeviceGuard guard(v->device);
    const size_t frame_bytes = (size_t)v->W * v->H * v->C;
    const int slot = (int)(v->step % 2);
    if (!v->d_frames[slot]) MIF_HIP(hipMalloc(&v->d_frames[slot], frame_bytes * v->batch));
    if (!v->upload_ev[slot]) MIF_HIP(hipEventCreateWithFlags(&v->upload_ev[slot], hipEventDisableTiming));
    // The caller may rewrite a buffer once the call after next has returned: this is that call for the
    // buffer handed over two calls ago (same slot), so its upload has to be over before we return.
    // A copy from pinned memory is asynchronous and nothing else on this path waits on the host.
    if (v->upload_pending[slot]) {
        MIF_HIP(hipEventSynchronize(v->upload_ev[slot]));
        v->upload_pending[slot] = false;
    }
    // the buffer was last read two batches ago, while this slot's handle built the first octave's scale
    // space; the upload then runs beside everything after that (asynchronous for pinned memory)
    int rc = mif_sift_wait_input_read(v->sift[slot], v->cstream);
    if (rc != MIF_OK) return rc;
    MIF_HIP(hipMemcpyAsync(v->d_frames[slot], frames, frame_bytes * nframes, hipMemcpyHostToDevice, v->cstream));
    MIF_HIP(hipEventRecord(v->upload_ev[slot], v->cstream));
    v->upload_pending[slot] = true;
    MIF_HIP(hipStreamWaitEvent(v->xstream, v->upload_ev[slot], 0));
    return submit(v, v->d_frames[slot], nframes);
}

int mif_video_collect(mif_video* v, mif_video_frame* out, int capacity, int* n)
{
    if (!v || !n || (!out && capacity > 0)) { mif::set_error("mif_video_collect: NULL argument"); return MIF_ERR_INVALID_ARGUMENT; }
    *n = 0;
    mif::DeviceGuard guard(v->device);
    if (v->pending) {
        const long long k = v->step - 1;
        int rc = enqueue_match(v, k, v->pending_frames);
        if (rc != MIF_OK) return rc;
        v->pending = false;
    }
    MIF_HIP(hipStreamSynchronize(v->xstream));
    MIF_HIP(hipStreamSynchronize(v->cstream));   // every upload is over: all host buffers are free again
    v->upload_pending[0] = v->upload_pending[1] = false;
    int total = 0;
    for (const auto& r : v->inflight) total += r.nframes;
    if (total > capacity) {
        mif::set_error("mif_video_collect: %d frames are waiting, capacity is %d", total, capacity);
        return MIF_ERR_CAPACITY;
    }
    int at = 0;
    for (auto& r : v->inflight) {
        for (int b = 0; b < r.nframes; b++, at++) {
            const int32_t* c = r.h_counts + (size_t)b * kCntWords;
            mif_video_frame& f = out[at];
            f.status = c[kCntStatus];
            for (int o = 0; o < MIF_SIFT_OCTAVES; o++) f.keypoints[o] = c[o];
            const bool matched = r.matched && (b > 0 || r.first_has_previous);
            f.good_matches = matched ? r.h_summary[2 * b] : -1;
            f.query_rows = matched ? r.h_summary[2 * b + 1] : c[kCntTotal];
        }
        v->pool.push_back(r);
    }
    v->inflight.clear();
    *n = at;
    return MIF_OK;
}

int mif_video_batch(const mif_video* v) { return v ? v->batch : MIF_ERR_INVALID_ARGUMENT; }

int mif_video_set_matching(mif_video* v, int enable)
{
    if (!v) { mif::set_error("mif_video_set_matching: NULL handle"); return MIF_ERR_INVALID_ARGUMENT; }
    v->matching = enable != 0;
    return MIF_OK;
}

} // extern "C"
