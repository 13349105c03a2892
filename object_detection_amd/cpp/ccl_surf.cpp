// ccl_surf:: host shims over the C ABI (see ccl_surf.h). One process-global
// pipeline like the reference's surf_items::alg_surf (ccl_surf.cpp:766).
#include "ccl_surf.h"

#include <cstring>
#include <fstream>
#include <iostream>

#include "../../include/mi_features.h"

static_assert(sizeof(ccl_surf::Ipoint) == sizeof(mif_ipoint), "Ipoint must match the C ABI record");

namespace
{
    struct Global
    {
        mif_surf* handle = nullptr;
        int rows = 0, cols = 0, channels = 0, device = 0;
        bool initialized = false;
        float last_ms = 0.0f;
        ~Global() { mif_surf_destroy(handle); }
    };

    Global& g()
    {
        static Global inst;
        return inst;
    }

    const char* const kKernelNames[] = { "createDescriptors", "getOrientationStep1", "getOrientationStep2",
                                         "hessian_det", "scan", "scan4", "scanImage", "transpose", "transposeImage",
                                         "non_max_supression", "normalizeDescriptors" };
}

namespace ccl_surf
{
    bool initialize_surf_driver(const std::vector<std::pair<std::string, std::string>>& kernel_path_name,
                                const bool dump, const float offload, const std::string,
                                const std::chrono::time_point<std::chrono::system_clock>*)
    {
        if (!(offload == -1.0f || (offload >= 0.0f && offload <= 1.0f))) return false;
        for (const auto& path_name : kernel_path_name) {
            bool known = false;
            for (const char* name : kKernelNames) known = known || path_name.second == name;
            if (!known) {
                std::cerr << "Expected kernel function names:" << std::endl;
                for (const char* name : kKernelNames) std::cerr << name << std::endl;
                return false;
            }
            if (dump) std::cout << "Build task:\t" << path_name.second << std::endl;
        }
        g().initialized = true;
        return true;
    }

    IpVec run(const mif::Image& img, bool)
    {
        Global& G = g();
        if (!G.initialized) {
            std::cerr << "Please initialize_surf_driver first .. exit!" << std::endl;
            return {};
        }
        if (img.empty()) {
            std::cerr << "Some err on main surf loop, FIXME!!" << std::endl;
            return {};
        }
        if (G.handle == nullptr || G.rows != img.rows || G.cols != img.cols || G.channels != img.channels()) {
            mif_surf_destroy(G.handle);
            G.handle = nullptr;
            if (mif_surf_create(G.device, img.cols, img.rows, img.channels(), &G.handle) != MIF_OK) {
                std::cerr << "Some err on main surf loop, FIXME!! " << mif_last_error() << std::endl;
                return {};
            }
            G.rows = img.rows; G.cols = img.cols; G.channels = img.channels();
        }
        IpVec pts((size_t)mif_surf_capacity(G.handle));
        int n = 0;
        const int rc = mif_surf_run(G.handle, img.data, reinterpret_cast<mif_ipoint*>(pts.data()), (int)pts.size(), &n,
                                    nullptr, &G.last_ms);
        if (rc != MIF_OK || n == 0) {
            std::cerr << "Some err on main surf loop, FIXME!!" << std::endl;
            return {};
        }
        pts.resize((size_t)n);
        return pts;
    }

    bool write_task_times_to_file(const std::string file)
    {
        if (file.empty()) return false;
        std::ofstream ofs(file);
        if (!ofs.is_open()) {
            std::cerr << "Couldn't open:\t" << file << " ,fixme!!!" << std::endl;
            return false;
        }
        ofs << "surf_frame_gpu_ms," << g().last_ms << std::endl;
        return true;
    }

    void set_device(int device)
    {
        Global& G = g();
        if (device == G.device) return;
        mif_surf_destroy(G.handle);
        G.handle = nullptr;
        G.device = device;
    }
}
