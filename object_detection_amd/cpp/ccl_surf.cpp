// ccl_surf:: host shims over the C ABI (see ccl_surf.h). Process-global state like the reference's
// surf_items::alg_surf (ccl_surf.cpp:766), which is rebuilt when the image grows (ccl_surf.cpp:806-810).
// Here the last few geometries keep their pipelines, so that a caller that alternates between two
// images (the -m demo) builds each once.
#include "ccl_surf.h"

#include <cstring>
#include <fstream>
#include <iostream>

#include "../../include/mi_features.h"

static_assert(sizeof(ccl_surf::Ipoint) == sizeof(mif_ipoint), "Ipoint must match the C ABI record");

namespace
{
    struct Slot
    {
        mif_surf* handle = nullptr;
        int rows = 0, cols = 0, channels = 0;
        unsigned long used = 0;
    };
    constexpr int kSlots = 4;

    struct Global
    {
        Slot slot[kSlots];
        unsigned long clock = 0;
        mif_surf* handle = nullptr;          // the pipeline of the last call (one of the slots')
        int device = 0;
        bool initialized = false, quiet = false;
        float last_ms = 0.0f;
        void drop_all()
        {
            for (Slot& sl : slot) {
                mif_surf_destroy(sl.handle);
                sl = Slot();
            }
            handle = nullptr;
        }
        ~Global() { drop_all(); }
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
        Slot* hit = nullptr;
        Slot* victim = &G.slot[0];
        for (Slot& sl : G.slot) {
            if (sl.handle && sl.rows == img.rows && sl.cols == img.cols && sl.channels == img.channels()) hit = &sl;
            if (!sl.handle || (victim->handle && sl.used < victim->used)) victim = &sl;
        }
        if (!hit) {
            mif_surf_destroy(victim->handle);
            *victim = Slot();
            G.handle = nullptr;
            if (mif_surf_create(G.device, img.cols, img.rows, img.channels(), &victim->handle) != MIF_OK) {
                std::cerr << "Some err on main surf loop, FIXME!! " << mif_last_error() << std::endl;
                return {};
            }
            victim->rows = img.rows; victim->cols = img.cols; victim->channels = img.channels();
            mif_surf_set_timing(victim->handle, 1);
            hit = victim;
        }
        hit->used = ++G.clock;
        G.handle = hit->handle;
        IpVec pts((size_t)mif_surf_capacity(G.handle));
        int n = 0;
        const int rc = mif_surf_run(G.handle, img.data, reinterpret_cast<mif_ipoint*>(pts.data()), (int)pts.size(), &n,
                                    nullptr, &G.last_ms);
        if (rc != MIF_OK || n == 0) {
            if (!G.quiet || rc != MIF_OK) std::cerr << "Some err on main surf loop, FIXME!!" << std::endl;
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
        // one record per stage of the last call (the reference: one per task, ccl_surf.cpp:859-878)
        ofs << "surf_frame_gpu_ms," << g().last_ms << std::endl;
        float ms[MIF_SURF_STAGES];
        if (g().handle && mif_surf_stage_times(g().handle, ms) == MIF_OK) {
            static const char* names[MIF_SURF_STAGES] = {"integral_image", "hessian_det", "non_max_supression",
                                                         "getOrientation", "createDescriptors"};
            for (int i = 0; i < MIF_SURF_STAGES; i++) ofs << names[i] << "_gpu_ms," << ms[i] << std::endl;
        }
        return true;
    }

    void set_device(int device)
    {
        Global& G = g();
        if (device == G.device) return;
        G.drop_all();
        G.device = device;
    }

    void set_quiet(bool quiet) { g().quiet = quiet; }
}
