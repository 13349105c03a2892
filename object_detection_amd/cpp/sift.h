// SIFT feature extraction, host interface. Same names, arguments and return
// codes as the reference's SIFT/sift.h:42-71; the work runs on one MI355X
// through libmi_features.so (include/mi_features.h).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace SIFT
{
    // SIFT/sift.h:42-49. x, y are in input-image pixels, sigma = 1.6 * 2^((level + offset) / 3),
    // scale_id is the pyramid level 1..3, octave_id 0..2.
    struct sfeature
    {
        float x;
        float y;
        float sigma;
        size_t scale_id;
        size_t octave_id;
    };

    // Detector only (SIFT/sift.h:51-58). `offload` was the CPU/GPU work split of the OpenCL driver;
    // it is checked like common/parse_args.h:89-95 (-1 or within [0, 1], else -2 is returned) and
    // otherwise ignored. Features are appended to `features` (the reference uses push_back,
    // SIFT/sift.cpp:1901). Returns 0, or a negative error code.
    int extract_features(
        size_t& elapsed_time,
        float offload,
        const unsigned char* input_color_image,
        int width, int height,
        int color_depth,
        std::vector<sfeature>& features,
        bool up_scale = false);

    // Detector + 128-d descriptors + one orientation per feature (SIFT/sift.h:60-69). Returns -1
    // and appends nothing when an octave finds no keypoint (SIFT/sift.cpp:1257-1258).
    int extract_features(
        size_t& elapsed_time,
        float offload,
        const unsigned char* input_color_image,
        int width, int height,
        int color_depth,
        std::vector<sfeature>& features,
        std::vector<float>& descriptors,
        std::vector<float>& orientations,
        bool up_scale = false);

    // SIFT/sift.h:71. Writes <path>sift_stage_times.csv with the GPU time per octave and stage of
    // the last frame, and per stage and octave the reference's three record files
    // <path><stage><octave>0_off.dat / _cpu.dat / _gpu.dat (SIFT/sift.cpp:915-957) with one line per
    // extraction since the pipeline was built; a stage stands where the reference has an OpenCL task.
    void store_task_exec_times(const std::string& path);

    // Probes of the reference's _DEBUG build (SIFT/sift.h:9-26): number of octaves / scales and
    // one image of the last extraction's scale space. The pointer stays valid until the next call.
    size_t cnt_octaves();
    size_t cnt_scales();
    const float* get_blur(float offload, const unsigned char* image, int width, int height, int color_depth,
                          int oid, int sid);
    const float* get_diff_blur(float offload, const unsigned char* image, int width, int height, int color_depth,
                               int oid, int sid);

    // Input of octave oid's scale space: the gray image (octave 0) or the decimated previous octave.
    const float* get_gray(float offload, const unsigned char* image, int width, int height, int color_depth,
                          int oid);

    // Resampling probes of the reference's _DEBUG build (SIFT/sift.h:28-38): nearest-neighbour resize of
    // the 8-bit image, resp. of its gray conversion, by (ratio_x, ratio_y). An empty vector reports an
    // error (the reference's versions do not report any).
    std::vector<std::uint8_t> test_kResize(float offload, const unsigned char* image, int width, int height,
                                           int color_depth, float ratio_x, float ratio_y);
    std::vector<float> test_ColorToGray_Resize(float offload, const unsigned char* image, int width, int height,
                                               int color_depth, float ratio_x, float ratio_y);

    // GPU used by the calls above (default 0); changing it drops the cached pipelines.
    void set_device(int device);
    // true: the calls above print none of the reference's progress lines ("Initialize ...", "Execute ...",
    // "Elapsed_time_ms,"): for timing loops. Errors are still reported.
    void set_quiet(bool quiet);
}
