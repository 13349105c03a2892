// SURF feature extraction, host interface. Same names and argument meaning as
// the reference's SURF/coopcl_surf/ccl_surf/ccl_surf.h:8-32; cv::Mat is replaced
// by mif::Image (same member names), the work runs on one MI355X through
// libmi_features.so.
#pragma once
#include <chrono>
#include <string>
#include <utility>
#include <vector>

#include "image_view.h"
#include "surf_constants.h"

namespace ccl_surf
{
    //! Interest point with its descriptor (ccl_surf.h:11-19)
    typedef struct {
        float x;
        float y;
        float scale;
        float orientation;
        int laplacian;
        int clusterIndex;
        float descriptor[64];
    } Ipoint;

    typedef std::vector<Ipoint> IpVec;

    // ccl_surf.h:23-28. The kernel source list, the dump switch, the JIT flags and the CPU/GPU
    // offload fraction configured the replaced OpenCL driver: the kernel names are still checked
    // against the reference's list (ccl_surf.cpp:13-16, 768-784), offload is range-checked, and
    // everything else is ignored. Returns false on a bad name or offload value.
    bool initialize_surf_driver(
        const std::vector<std::pair<std::string, std::string>>& kernel_path_name,
        const bool dump = false,
        const float offload = 0.0f,
        const std::string jit_flags = "",
        const std::chrono::time_point<std::chrono::system_clock>* app_begin = { nullptr });

    // ccl_surf.h:30. Empty vector if the driver was not initialised, on any error, and when no
    // interest point is found (ccl_surf.cpp:791-795, 813-817).
    IpVec run(const mif::Image& img, bool use_images_in_kernelfunc = false);

    // ccl_surf.h:32. Writes the GPU time of the last frame and of each of its stages to `file`.
    bool write_task_times_to_file(const std::string file);

    void set_device(int device);
    // true: no progress or "no point found" lines (timing loops); errors are still reported.
    void set_quiet(bool quiet);
}
