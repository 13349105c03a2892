/*
 * mi_features.h -- C ABI of libmi_features.so: GPU-resident SIFT / SURF feature
 * extraction and brute-force descriptor matching for AMD MI355X (gfx950).
 *
 * This is the boundary a maintainer of the reference binds instead of the
 * coopcl OpenCL driver. Each entry point names the reference interface it
 * replaces. All functions return 0 on success and a negative MIF_ERR_* code on
 * failure; they never call exit(). mif_last_error() returns a description of
 * the last failure on the calling thread.
 *
 * Pointers whose name starts with d_ are device (HBM) addresses on the handle's
 * GPU; all others are host addresses. No type of this header depends on HIP or
 * torch.
 */
#ifndef MI_FEATURES_H
#define MI_FEATURES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIF_OK 0
/* An octave produced no keypoint; SIFT::extract_features returns -1 in that
 * case (reference SIFT/sift.cpp:1257-1258, 1870-1876). */
#define MIF_ERR_NO_KEYPOINTS (-1)
#define MIF_ERR_INVALID_ARGUMENT (-2)
#define MIF_ERR_HIP (-3)
#define MIF_ERR_CAPACITY (-4)
#define MIF_ERR_UNSUPPORTED (-5)

#define MIF_SIFT_OCTAVES 3   /* _SIFT_MAX_OCTAVES, SIFT/sift.cpp:24 */
#define MIF_SIFT_LEVELS 6    /* _SIFT_INTVLS + 3,  SIFT/sift.cpp:17,23 */
#define MIF_SIFT_DESC_DIM 128
/* Timed stages of an octave: 0 base image (gray conversion or decimation),
 * 1 scale space (the six blur + DoG launches), 2 detector (detect, scan, emit),
 * 3 orientation + descriptor. */
#define MIF_SIFT_STAGES 4

const char* mif_last_error(void);
int mif_device_count(void);
/* Test hook without counterpart in the reference: evaluates on the GPU the elementary functions the
 * kernels use in place of the reference's native_exp / native_powr / sin / cos / atan2 (OpenCL built-ins
 * under -cl-fast-relaxed-math, SIFT/sift.cpp:1369-1370), so that a checker compiled for the host from
 * the same definitions can be compared with them bit for bit. op: 0 exp, 1 exp2, 2 sin, 3 cos,
 * 4 atan2(a, b), 5 exp without range guards (-87 <= a <= 88). a, b, out: n host floats (b only for op 4). */
int mif_math_probe(int device, int op, const float* a, const float* b, float* out, int n);

/* ------------------------------------------------------------------ SIFT -- */

typedef struct mif_sift mif_sift;

/* Replaces SIFT::init / sift_algorithm::sift_algorithm / sift_octave ctor
 * (SIFT/sift.cpp:2148-2189, 1731-1770, 1291-1538): builds the per-octave buffer
 * plan for a width x height image with `channels` = 1 (gray) or 3 (interleaved
 * BGR) on GPU `device`. up_scale != 0 builds the first octave at twice the input size by
 * nearest-neighbour resampling (kResize_NN_*, SIFT/sift.cpp:134-210, 1140-1177, 1306-1310); keypoint
 * coordinates are then in pixels of the resampled image, as in the reference. */
int mif_sift_create(int device, int width, int height, int channels, int up_scale, mif_sift** out);
/* Same, with room for `batch` frames per enqueue: every kernel launch then
 * covers the whole batch (frames are independent, as successive calls of the
 * reference are). mif_sift_create is batch = 1. */
int mif_sift_create_batched(int device, int width, int height, int channels, int up_scale, int batch,
                            mif_sift** out);
int mif_sift_batch(const mif_sift* s);
int mif_sift_destroy(mif_sift* s);

/* Geometry of octave o: size_t(w / 2^o), size_t(h / 2^o), max_features =
 * items / 100 (SIFT/sift.cpp:1313-1314, 1356). Any out pointer may be NULL. */
int mif_sift_octave_info(const mif_sift* s, int octave, int* width, int* height, int* max_features);
/* Sum of max_features over the octaves: the row capacity of every result buffer. */
int mif_sift_capacity(const mif_sift* s);

/* Replaces both SIFT::extract_features overloads (SIFT/sift.h:51-69,
 * SIFT/sift.cpp:2191-2217) for host buffers.
 *   pixels        width*height*channels bytes
 *   keypoints     capacity x 4 floats {x, y, sigma, level}   (float4 of sift.cpp:410-413)
 *   orientations  capacity floats, or NULL (detector only)
 *   descriptors   capacity x 128 floats, or NULL
 *   counts        keypoints per octave; rows are octave-major (sift.cpp:1889-1911)
 *                 and inside an octave ordered by (level, y, x) of the detecting pixel
 *   elapsed_ms    GPU time from first to last kernel of the frame, or NULL
 * Returns MIF_ERR_NO_KEYPOINTS (and all counts 0) if descriptors were requested
 * and some octave found nothing. */
int mif_sift_extract(mif_sift* s, const uint8_t* pixels, float* keypoints, float* orientations,
                     float* descriptors, int capacity, int counts[MIF_SIFT_OCTAVES], float* elapsed_ms);

/* Device-resident form of the same call: d_pixels holds nframes (<= batch)
 * frames back to back in HBM; they are enqueued on the handle's stream and
 * nothing is copied back. want_descriptors = 0 stops after the detector (first
 * overload of sift.h). */
int mif_sift_enqueue(mif_sift* s, const void* d_pixels, int nframes, int want_descriptors);
/* The two phases of mif_sift_enqueue as separate calls, for callers that schedule other work of
 * their own between or beside them: _pyramid builds the base images and scale spaces (+ DoG) of all
 * octaves (the reference's kColor / kDown / kBlurH / kBlurV / kDiff launches, SIFT/sift.cpp:1140-1230,
 * bandwidth kernels), _features runs detection, orientations and descriptors on them (kDetector,
 * kOrientation, kDescriptor, kNormalize, sift.cpp:1232-1282). mif_sift_enqueue is _pyramid followed
 * by _features. */
int mif_sift_enqueue_pyramid(mif_sift* s, const void* d_pixels, int nframes);
int mif_sift_enqueue_features(mif_sift* s, int want_descriptors);
/* Copies the descriptor rows and the keypoint count of one frame of the last batch into caller-owned
 * device memory (d_descriptors: room for [capacity][128] floats, 16-byte aligned; the frame's rows are
 * written, the rest is left as it is), in stream order on the handle's stream: the frame a video
 * caller matches the next batch's first frame against (SIFT/main.cpp:103-153 keeps the previous
 * frame's features on the host). */
int mif_sift_snapshot_frame(mif_sift* s, int frame, float* d_descriptors, int32_t* d_count);
/* The readback of SIFT/sift.cpp:1889-1911 (features(), orientations() and descriptors() appended octave
 * by octave to the caller's vectors) for a whole batch: packs the rows of the first nframes frames of the
 * last enqueued batch back to back -- frame after frame, every frame octave-major as mif_sift_extract
 * returns it -- into caller-owned device arrays, in stream order on the handle's stream, so that a
 * batch's results leave the GPU with one copy per array instead of one per frame.
 *   d_keypoints     max_rows x 4 floats, 16-byte aligned
 *   d_orientations  max_rows floats, or NULL together with d_descriptors (detector only)
 *   d_descriptors   max_rows x 128 floats, 16-byte aligned
 *   d_offsets       int32[nframes + 2]: [f] = first packed row of frame f, [nframes] = rows packed,
 *                   [nframes + 1] = rows the frames hold (larger than [nframes] when max_rows cut the
 *                   batch short: the surplus rows are dropped from the end) */
int mif_sift_pack_results(mif_sift* s, int nframes, float* d_keypoints, float* d_orientations, float* d_descriptors,
                          int32_t* d_offsets, int max_rows);
/* Orders two streams of the caller's: everything enqueued on `waiter` after this call runs after
 * everything enqueued on `signaller` before it. No counterpart in the reference, whose driver
 * chains tasks with add_dependence (clDriver.h); streams are HIP stream handles (mif_sift_stream). */
int mif_stream_wait_stream(int device, void* waiter, void* signaller);

/* A stream whose kernels run only on the compute units whose bit is set in cu_mask (words x 32 bits,
 * bit i = compute unit i in the driver's numbering, which deals consecutive bits to the device's XCDs in
 * turn: the low n bits are n / 8 compute units of every XCD). Two streams with disjoint masks run their
 * kernels side by side, each on its share of the device. No counterpart in the reference (its two devices
 * are a CPU and a GPU, clDriver.h:1311-1432). The mask must be one contiguous run of at least 8 set
 * bits (so that every XCD keeps a compute unit) below the device's compute-unit count; anything else is
 * refused with MIF_ERR_INVALID_ARGUMENT before the runtime is called (a mask of every other bit crashed
 * ROCm 7.2 inside hipExtStreamCreateWithCUMask). mif_stream_destroy waits for the stream's work first. */
int mif_stream_create_masked(int device, const uint32_t* cu_mask, int words, void** stream);
int mif_stream_destroy(int device, void* stream);

/* Blocks until everything enqueued on the handle has finished. */
int mif_sift_sync(mif_sift* s);
/* Makes everything enqueued on s from now on wait for what has been enqueued
 * on `other` so far (an event dependency between the two handles' streams; no
 * host wait). Used when one handle's results feed work ordered on another. */
int mif_sift_wait_for(mif_sift* s, mif_sift* other);
/* Device addresses of the last batch's results, valid until the next enqueue.
 * Frame f of the batch starts at row f * capacity of keypoints
 * float4[batch][capacity], orientations float[batch][capacity], descriptors
 * float[batch][capacity][128] and at counts + 8 f of counts int32[batch][8] =
 * {n_oct0, n_oct1, n_oct2, n_total, raw_oct0, raw_oct1, raw_oct2, status}.
 * Any out pointer may be NULL. */
int mif_sift_device_results(mif_sift* s, void** d_keypoints, void** d_orientations, void** d_descriptors,
                            void** d_counts);
/* The handle's hipStream_t, for callers that order their own work after a frame. */
void* mif_sift_stream(mif_sift* s);
/* Makes the handle enqueue on a stream of the caller's from now on (NULL: back to its own). Handles
 * whose work is ordered one after the other anyway -- the two handles a video pipeline alternates
 * between -- can share one stream: a process has few hardware queues (4 by default) for all its
 * streams, and streams that share a queue wait behind each other's kernels, e.g. an upload behind a
 * matcher. Waits for what the handle has enqueued so far. The stream stays the caller's: it must
 * outlive the handle, or be detached with mif_sift_set_stream(s, NULL) before it is destroyed (every
 * later call on the handle enqueues on it or waits for it). No counterpart in the reference. */
int mif_sift_set_stream(mif_sift* s, void* stream);
/* Makes everything enqueued on `waiter` (a hipStream_t) from now on wait until the pixels of the batch
 * last enqueued on s have been read (they are read while the first octave's scale space is built; the
 * rest of the batch's work does not touch them): the point from which the caller may overwrite a
 * device-side frame buffer, e.g. with the upload of the batch after next. No host wait; returns at
 * once when nothing has been enqueued. The reference copies the caller's pixels synchronously
 * (update_pixels, SIFT/sift.cpp:1540-1552). */
int mif_sift_wait_input_read(mif_sift* s, void* waiter);
/* Waits for the stream and reads frame `frame`'s per-octave counts (clamped to
 * max_features) and raw detector totals. Either array may be NULL. Returns the
 * frame's status (0 or MIF_ERR_NO_KEYPOINTS). */
int mif_sift_read_counts(mif_sift* s, int frame, int counts[MIF_SIFT_OCTAVES], int raw_counts[MIF_SIFT_OCTAVES]);
/* Waits for the stream and copies one frame of the last batch to host buffers
 * laid out as for mif_sift_extract (orientations / descriptors may be NULL).
 * Returns the frame's status. */
int mif_sift_copy_results(mif_sift* s, int frame, float* keypoints, float* orientations, float* descriptors,
                          int capacity, int counts[MIF_SIFT_OCTAVES]);

/* Replace the _DEBUG probes SIFT::get_blur / get_diff_blur (SIFT/sift.h:13-21):
 * copy one image of the scale space of frame `frame` of the last batch to the
 * host. level 0..5 of the Gaussian pyramid, index 0..4 of the DoG stack. */
int mif_sift_copy_level(mif_sift* s, int frame, int octave, int level, float* out);
int mif_sift_copy_dog(mif_sift* s, int frame, int octave, int index, float* out);
/* Replaces the _DEBUG probe SIFT::get_gray (SIFT/sift.h:23-26, sift.cpp:1590-1595, 1773-1801): the
 * input of an octave's scale space before any blur, i.e. the gray conversion of the frame (resampled
 * when up_scale is set) for octave 0 and the decimated level 3 of the previous octave otherwise. The
 * reference hands out pyramid level 0, which its blur overwrites in place. Here the extraction
 * does not store that image (the first blur forms it on the fly), so this call forms it from the
 * pixels passed to the last extraction call on `s`, which must still be valid when it is a
 * caller-owned device buffer (mif_sift_enqueue*), resp. from the previous octave's scale space. */
int mif_sift_copy_base(mif_sift* s, int frame, int octave, float* out);
/* The resampling probes of the reference's debug build. mif_probe_resize_u8 replaces SIFT::test_kResize
 * (SIFT/sift.h:28-32, SIFT/sift.cpp:2097-2117): nearest-neighbour resampling of an 8-bit image of
 * 1 or 3 interleaved channels by (ratio_x, ratio_y) (kResize_NN_1ch_8b / kResize_NN_3ch_8b,
 * sift.cpp:159-206); `out` takes (int)(width * ratio_x) x (int)(height * ratio_y) pixels.
 * mif_probe_gray_resize replaces SIFT::test_ColorToGray_Resize (sift.h:34-38, sift.cpp:2119-2144):
 * kColor_interleaved_ch on a BGR image (channels must be 3), then kResize_NN_1ch_32b (sift.cpp:136-157);
 * `out` takes as many floats. Destination pixels whose source position lies outside the image are 0.
 * Like the reference's probes they set up and tear down their own device buffers per call. */
int mif_probe_resize_u8(int device, const uint8_t* pixels, int width, int height, int channels, float ratio_x,
                        float ratio_y, uint8_t* out);
int mif_probe_gray_resize(int device, const uint8_t* pixels, int width, int height, int channels, float ratio_x,
                          float ratio_y, float* out);
/* Gaussian taps of a level as the blur kernels use them (2*radius+1 floats;
 * Build_separable_filter_cof, SIFT/sift.cpp:819-889). Returns the radius. */
int mif_sift_taps(const mif_sift* s, int level, float* out, int capacity);
/* Per-stage GPU timing, the counterpart of the reference's per-task profile
 * dump (SIFT::store_task_exec_times, SIFT/sift.h:71, SIFT/sift.cpp:1674-1703).
 * While enabled, every enqueue records hipEvents on the handle's stream at the
 * stage boundaries of each octave; mif_sift_stage_times waits for the stream and
 * returns the last batch's milliseconds per [octave][stage]. */
int mif_sift_set_timing(mif_sift* s, int enable);
int mif_sift_stage_times(mif_sift* s, float ms[MIF_SIFT_OCTAVES][MIF_SIFT_STAGES]);
/* Kernel launches that built the scale spaces of the last batch enqueued on s, over all octaves: 18
 * when every level has a launch of its own (the reference: one kBlurH + one kBlurV task per level and
 * octave, SIFT/sift.cpp:1196-1218), fewer when two levels share one. For per-launch averages of the
 * scale-space stage time. */
int mif_sift_blur_launches(const mif_sift* s);
/* GPU time of the last synchronised frame in ms (hipEvent pair). */
int mif_sift_last_elapsed_ms(mif_sift* s, float* ms);

/* ------------------------------------------------------------------ SURF -- */

typedef struct mif_surf mif_surf;

/* Interest point with its 64-d descriptor: the layout of ccl_surf::Ipoint
 * (SURF/coopcl_surf/ccl_surf/ccl_surf.h:11-19). */
typedef struct mif_ipoint {
    float x;
    float y;
    float scale;
    float orientation;
    int32_t laplacian;
    int32_t clusterIndex;
    float descriptor[64];
} mif_ipoint;

#define MIF_SURF_LAYERS 8        /* response layers for OCTAVES = 3 (ccl_surf.cpp:115-153) */
#define MIF_SURF_MAX_POINTS 8000 /* INIT_FEATS, surf_constants.h */

/* Replaces ccl_surf::initialize_surf_driver + the Surf / FastHessian
 * constructors (ccl_surf.cpp:822-857, 604-663, 181-214) for a width x height
 * image with 1 (gray) or 3 (interleaved BGR) 8-bit channels. */
int mif_surf_create(int device, int width, int height, int channels, mif_surf** out);
/* Same, with room for `batch` frames per enqueue: every kernel launch then covers the whole batch
 * (frames are independent, as successive calls of ccl_surf::run are). mif_surf_create is batch = 1. */
int mif_surf_create_batched(int device, int width, int height, int channels, int batch, mif_surf** out);
int mif_surf_batch(const mif_surf* s);
int mif_surf_destroy(mif_surf* s);
int mif_surf_capacity(const mif_surf* s);

/* Replaces ccl_surf::run (ccl_surf.h:30, ccl_surf.cpp:789-820) for host
 * buffers: writes *count interest points in the order (non_max_supression
 * launch, row, column). The reference returns an empty vector when no point is
 * found; here that is *count = 0 with return value 0. raw_count (may be NULL)
 * receives the number of accepted positions before the cut at 8000. */
int mif_surf_run(mif_surf* s, const uint8_t* pixels, mif_ipoint* points, int capacity, int* count, int* raw_count,
                 float* elapsed_ms);

/* Device-resident form: d_pixels lives in HBM, work is enqueued on the handle's
 * stream, results stay on the device: pos float2[8000], scale float[8000],
 * laplacian int32[8000], orientation float[8000], descriptors float[8000][64],
 * counts int32[2] = {points, raw}. Any out pointer may be NULL.
 * Batches of 192 bands of 64 image rows and more build their integral images in one launch whose bands
 * wait for each other through memory; the wait is bounded, and a batch in which a band gave up is built
 * again by the launch-per-pass kernels enqueued behind it (same results, later). */
int mif_surf_enqueue(mif_surf* s, const void* d_pixels);
/* nframes (<= batch) frames back to back in HBM; frame f's results start at row f * 8000 of the point
 * arrays of mif_surf_device_results and at counts + 2 f. */
int mif_surf_enqueue_batch(mif_surf* s, const void* d_pixels, int nframes);
/* Waits for the stream and copies the interest points of one frame of the last batch to the host, laid
 * out as for mif_surf_run. */
int mif_surf_copy_points(mif_surf* s, int frame, mif_ipoint* points, int capacity, int* count, int* raw_count);
int mif_surf_sync(mif_surf* s);
/* Waits for the stream; *count = batches of this handle whose integral images had to be built again (see
 * mif_surf_enqueue). No counterpart in the reference: a diagnostic, expected to stay 0. */
int mif_surf_integral_fallbacks(mif_surf* s, int* count);
void* mif_surf_stream(mif_surf* s);
int mif_surf_device_results(mif_surf* s, void** d_pos, void** d_scale, void** d_laplacian, void** d_orientation,
                            void** d_descriptors, void** d_counts);

/* Per-stage GPU timing, the counterpart of ccl_surf::write_task_times_to_file (ccl_surf.h:32,
 * ccl_surf.cpp:859-878: one record per task). While enabled, every enqueue records hipEvents on the
 * handle's stream at the stage boundaries; mif_surf_stage_times waits for the stream and returns the last
 * batch's milliseconds per stage: 0 integral image (scan, transpose, scan, transpose in the reference),
 * 1 the eight hessian_det layers, 2 non_max_supression and the point list, 3 getOrientationStep1/2,
 * 4 createDescriptors + normalizeDescriptors and the packing of the Ipoint records. */
#define MIF_SURF_STAGES 5
int mif_surf_set_timing(mif_surf* s, int enable);
int mif_surf_stage_times(mif_surf* s, float ms[MIF_SURF_STAGES]);

/* Probes for tests: the integral image (width*height floats) and one response
 * layer (responses / laplacians, either may be NULL) of the last frame. */
int mif_surf_copy_integral(mif_surf* s, float* out);
int mif_surf_copy_integral_frame(mif_surf* s, int frame, float* out);
int mif_surf_layer_info(const mif_surf* s, int layer, int* width, int* height, int* step, int* filter);
int mif_surf_copy_layer(mif_surf* s, int layer, float* responses, int32_t* laplacians);

/* -------------------------------------------------------------- matching -- */

typedef struct mif_matcher mif_matcher;

/* Workspace for matching up to max_query x max_database descriptors on `device`. */
int mif_match_create(int device, int max_query, int max_database, mif_matcher** out);
/* Same, for up to max_pairs independent (query set, database) pairs per call. */
int mif_match_create_batched(int device, int max_query, int max_database, int max_pairs, mif_matcher** out);
int mif_match_destroy(mif_matcher* m);
/* enable = 0: every call of this matcher runs the plain fp32 search; default 1: large calls are
 * screened in bf16 first (see mif_match_top2_batch_device). The results are the same bits either
 * way; the switch exists to measure the two forms side by side. No counterpart in the reference. */
int mif_match_set_screening(mif_matcher* m, int enable);

/* How many times the screening of the matcher's last call contracted the queries with the whole
 * database in bf16: 0 (searched in fp32 throughout), 2, or 1 + 1/n when the first pass looked at every
 * n-th chunk of rows only (large searches). For throughput accounting; no counterpart in the reference. */
int mif_match_last_screen_passes(mif_matcher* m, double* passes);

/* Exact brute-force 2-nearest-neighbour search in L2; replaces the
 * cv::FlannBasedMatcher::match / knnMatch(k=2) calls of common/kpmatch.h:174-176
 * and 284-286. Row i of q / db starts at q + i*ld; only the first `dim` floats of
 * a row take part (dim = 127, ld = 128 reproduces the shifted descriptor row of
 * kpmatch.h:15,127-128,141-142). For every query row the two database rows with
 * the smallest distance are returned, nearest first, ties to the lowest index:
 * idx2[2*i], idx2[2*i+1] (-1 when the database has fewer rows) and the Euclidean
 * distances dist2[2*i], dist2[2*i+1] (+inf when missing). */
int mif_match_top2(mif_matcher* m, const float* q, int nq, const float* db, int nd, int dim, int ld,
                   int32_t* idx2, float* dist2);
/* Same, and also returns the ranking scores of the two winners (squared-distance
 * estimates |q|^2 + |d|^2 - 2 q.d, +inf when missing). Candidates of several
 * database shards are merged on (score, index), which reproduces the result of
 * one search over the whole database. score2 may be NULL. */
int mif_match_top2_scores(mif_matcher* m, const float* q, int nq, const float* db, int nd, int dim, int ld,
                          int32_t* idx2, float* dist2, float* score2);
/* Device-resident form, enqueued on `stream` (a hipStream_t, NULL = default).
 * d_nq / d_nd, when not NULL, point to device int32 row counts that override
 * nq / nd (which then are the capacities the grid is sized for). */
int mif_match_top2_device(mif_matcher* m, const float* d_q, int nq, const int32_t* d_nq, const float* d_db,
                          int nd, const int32_t* d_nd, int dim, int ld, int32_t* d_idx2, float* d_dist2,
                          void* stream);

/* The same search as mif_match_top2_device for callers that cannot allocate: tight serving loops,
 * harnesses that forbid hipMalloc, code that is about to be captured into a HIP graph (capture itself
 * has not been tested; make one call outside the capture first, which looks up what the later calls
 * reuse). No counterpart in the reference.
 * There is no handle: the workspace (pair table, work counters, row norms, per-split partial winners)
 * is one fixed array of 4.3 MB in device memory that every GPU the library runs on holds until the
 * process ends, so a call is limited to MIF_NOALLOC_MAX_QUERY query rows and
 * MIF_NOALLOC_MAX_DATABASE database rows (MIF_ERR_CAPACITY beyond; use a mif_matcher then). The call
 * enqueues kernels and one memset on `stream` (a hipStream_t of `device`, NULL = default) and calls no
 * hipMalloc / hipFree or allocating library routine; the first call on a device looks the array's
 * address and the device's compute-unit count up, later calls query nothing. Always the plain fp32 search; results are the bits mif_match_top2_device returns.
 * All calls on one GPU share the array: calls that may overlap on one GPU (different streams, different
 * threads) race and must be serialised by the caller. */
#define MIF_NOALLOC_MAX_QUERY 4096
#define MIF_NOALLOC_MAX_DATABASE 65536
int mif_match_top2_device_noalloc(int device, const float* d_q, int nq, const float* d_db, int nd, int dim, int ld,
                                  int32_t* d_idx2, float* d_dist2, void* stream);

/* One (query set, database) pair of a batched match; every pointer is a device
 * address. d_nq / d_nd may be NULL (the call's nq / nd are then the row counts).
 * d_good / d_summary / d_minmax are only written when a filter is requested and
 * may otherwise be NULL. */
typedef struct mif_match_pair {
    const float* d_q;
    const float* d_db;
    const int32_t* d_nq;
    const int32_t* d_nd;
    int32_t* d_idx2;     /* [nq][2] */
    float* d_dist2;      /* [nq][2] */
    uint8_t* d_good;     /* [nq] 1 = kept by the filter */
    int32_t* d_summary;  /* {kept, query rows} */
    float* d_minmax;     /* {min_dist, max_dist} of the nearest distances (filter 1) */
    float* d_score2;     /* [nq][2] ranking scores |q|^2 + |d|^2 - 2 q.d of the two winners, or NULL */
} mif_match_pair;

#define MIF_FILTER_NONE 0
/* Good-match rule of the SIFT overload of Plot_Img_Matches_ocv
 * (common/kpmatch.h:177-202): keep dist <= max(2.5 * min_dist, 0.35). */
#define MIF_FILTER_SIFT 1
/* Lowe ratio test of the SURF overload (common/kpmatch.h:289-297): keep d0 < 0.75 * d1. */
#define MIF_FILTER_RATIO 2

/* npairs independent top-2 searches in one set of launches on `stream`; all
 * pairs share the capacities nq / nd, dim and ld. `pairs` is a host array; it is
 * copied before the call returns (the kernels read a device copy of it). Calls
 * with at least 3.2e7 scores to evaluate (npairs * nq * nd; capacities count as an
 * eighth of themselves when the row counts live on the device) and at least 128
 * database rows are first screened with bf16 matrix instructions, database splits
 * of at most 32768 rows each, and only the rows that can still win are ranked in
 * fp32; a database too large for 64 such splits is searched in fp32 throughout.
 * Either way the results are those of the plain fp32 search, bit for bit.
 * A mif_matcher owns one set of work counters and candidate buffers: use it from
 * one stream at a time (calls on the same stream queue up behind each other; calls
 * on different streams that may overlap need a matcher each). */
int mif_match_top2_batch_device(mif_matcher* m, const mif_match_pair* pairs, int npairs, int nq, int nd, int dim,
                                int ld, int filter_mode, void* stream);

/* --------------------------------------------- database sharded over GPUs -- */

/* A descriptor database split row-wise over several GPUs. Every GPU searches its rows
 * (mif_match_top2_batch_device), the per-shard winners (ranking score, global row index, distance:
 * 24 bytes per query and shard) are exchanged with ONE RCCL all-gather over xGMI, and every GPU merges
 * them on (score, index), so that every rank ends up with the result of one search over the whole
 * database, bit for bit. The reference has no counterpart (its matcher is the single FLANN call of
 * common/kpmatch.h:174-176, 284-286; its only split is the CPU/GPU NDRange split of
 * clDriver.h:1879-1921); this serves the same call sites for databases that do not fit, or should not
 * wait for, one GPU.
 *
 * A handle drives `nlocal` consecutive ranks first_rank .. first_rank + nlocal - 1 of `nranks` from the
 * calling process, rank first_rank + i on GPU devices[i]:
 *   - one process per GPU: nlocal = 1; one rank calls mif_match_sharded_unique_id, hands the
 *     MIF_SHARD_ID_BYTES bytes to the others by its own means (MPI, a socket, torch.distributed) and all
 *     call mif_match_sharded_create with it (a collective call: it returns when all ranks have joined);
 *   - one process for all GPUs of a node: nlocal = nranks, id may be NULL; the RCCL calls of the shards
 *     are fused in one group.
 * Shards that share a GPU (only when nlocal = nranks; RCCL cannot place two ranks on one GPU) exchange
 * their messages with device-to-device copies instead of RCCL: the form for machines with one GPU.
 * RCCL is loaded (dlopen of librccl.so.1) by the first of these calls that needs it; MIF_ERR_UNSUPPORTED
 * when it is not installed. */
typedef struct mif_match_sharded mif_match_sharded;
#define MIF_SHARD_ID_BYTES 128
int mif_match_sharded_unique_id(void* id /* [MIF_SHARD_ID_BYTES] */);
int mif_match_sharded_create(const int* devices, int nlocal, int first_rank, int nranks, const void* id,
                             int max_query, int max_shard_rows, mif_match_sharded** out);
int mif_match_sharded_destroy(mif_match_sharded* h);
/* nlocal, nranks and whether the exchange runs over RCCL (1) or device copies (0); any pointer may be NULL. */
int mif_match_sharded_ranks(const mif_match_sharded* h, int* nlocal, int* nranks, int* uses_rccl);
/* mif_match_set_screening for every shard's matcher. */
int mif_match_sharded_set_screening(mif_match_sharded* h, int enable);
/* mif_match_last_screen_passes of the first local shard's matcher (all shards choose alike). */
int mif_match_sharded_last_screen_passes(mif_match_sharded* h, double* passes);

/* What one local rank brings to a search; all pointers are device addresses on that rank's GPU. */
typedef struct mif_shard_io {
    const float* d_q;      /* [nq][ld] the queries (the same rows on every rank) */
    const float* d_shard;  /* [shard_rows][ld] this rank's database rows */
    int32_t shard_rows;
    int32_t global_offset; /* index of the shard's first row in the whole database */
    int32_t* d_idx2;       /* [nq][2] out: GLOBAL row indices of the two nearest rows, -1 when missing */
    float* d_dist2;        /* [nq][2] out: their Euclidean distances, +inf when missing */
    void* stream;          /* hipStream_t to enqueue on; NULL = a stream the handle owns for this rank.
                            * Keep one stream per rank from call to call: the handle's buffers are reused in
                            * stream order. mif_match_sharded_sync and _destroy wait for the end of the last
                            * search on whichever stream it ran. */
} mif_shard_io;

/* One search: io[i] belongs to local rank i. nq, dim and ld must be the same on every rank of the
 * communicator and every rank must make the call (it contains a collective). Everything is enqueued;
 * nothing waits on the host. */
int mif_match_sharded_search(mif_match_sharded* h, const mif_shard_io* io, int nq, int dim, int ld);
int mif_match_sharded_sync(mif_match_sharded* h);

/* ------------------------------------------------------- frame sequences -- */

/* The -v loop of SIFT/main.cpp:103-153 (one extract_features call per frame, the previous
 * frame's features kept for comparison) for batches of frames that stay on the GPU: every
 * frame is extracted and matched against its predecessor (descriptor rows as
 * common/kpmatch.h:127-142 builds them, nearest neighbour kept under the good-match rule
 * of kpmatch.h:177-202). Two batched SIFT handles alternate; they and the matcher share one
 * stream (scale spaces of batch k, matcher of batch k - 1, features of batch k), uploads of
 * host frames run beside it on a second one; nothing waits on the host before
 * mif_video_collect. Frame 0 of a batch is matched against the last frame of the batch
 * before it. */
typedef struct mif_video mif_video;
typedef struct mif_video_frame {
    int32_t status;                      /* 0, or MIF_ERR_NO_KEYPOINTS (SIFT/sift.cpp:1257-1258) */
    int32_t keypoints[MIF_SIFT_OCTAVES]; /* per octave, clamped to the list limit */
    int32_t good_matches;                /* matches kept against the previous frame; -1 for the very first frame */
    int32_t query_rows;                  /* descriptor rows of this frame */
} mif_video_frame;

int mif_video_create(int device, int width, int height, int channels, int batch, mif_video** out);
int mif_video_destroy(mif_video* v);
int mif_video_batch(const mif_video* v);
/* enable = 0: extraction only (batches submitted from now on are not matched and report
 * good_matches = -1), e.g. for a batch of unrelated images. Default: matching on. */
int mif_video_set_matching(mif_video* v, int enable);
/* Enqueues the extraction of nframes (<= batch) frames, u8 [nframes][height][width][channels],
 * and the matching of the batch submitted before. _device: the frames are in device memory and
 * must stay untouched until the next but one submit or a collect. _host: the frames are uploaded
 * on a copy stream beside the previous batch's work (asynchronously when the memory is pinned);
 * the buffer may be reused once the call after next has returned, or after a collect. */
int mif_video_submit_device(mif_video* v, const void* d_frames, int nframes);
int mif_video_submit_host(mif_video* v, const uint8_t* frames, int nframes);
/* Waits for everything submitted and returns one record per frame since the last collect, in
 * submission order. MIF_ERR_CAPACITY (nothing consumed) when more than `capacity` are waiting. */
int mif_video_collect(mif_video* v, mif_video_frame* out, int capacity, int* n);

#ifdef __cplusplus
}
#endif
#endif /* MI_FEATURES_H */
