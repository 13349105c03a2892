// SIFT:: host shims over the C ABI (see sift.h). Process-global state like the reference's
// `static std::unique_ptr<sift_algorithm> sift` (SIFT/sift.cpp:1992, 2148-2189), which is rebuilt
// whenever the image geometry changes. Here the last few geometries keep their pipelines, so that a
// caller that alternates between two images (the -m demo: object, scene, object, ...) builds each once.
#include "sift.h"

#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>

#include "../../include/mi_features.h"

namespace
{
    struct Slot
    {
        mif_sift* handle = nullptr;
        int width = 0, height = 0, depth = 0;
        bool up = false;
        unsigned long used = 0;
    };
    constexpr int kSlots = 4;

    struct Global
    {
        Slot slot[kSlots];
        unsigned long clock = 0;
        mif_sift* handle = nullptr;          // the pipeline of the last call (one of the slots')
        int device = 0;
        std::vector<float> kp, ori, desc, probe;
        // per-stage GPU times of every extraction so far, [call][octave][stage] (the reference keeps one
        // record per task call, clDriver.h:1172-1247)
        std::vector<float> history;
        bool quiet = false;                  // no progress lines (timing loops)
        void drop_all()
        {
            for (Slot& sl : slot) {
                mif_sift_destroy(sl.handle);
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

    bool offload_ok(float offload) { return offload == -1.0f || (offload >= 0.0f && offload <= 1.0f); }

    // SIFT::init, SIFT/sift.cpp:2148-2189
    int init(const unsigned char* image, int width, int height, int depth, bool up_scale)
    {
        if (image == nullptr) {
            std::cerr << "Load image failed, fixme!" << std::endl;
            return MIF_ERR_INVALID_ARGUMENT;
        }
        Global& G = g();
        Slot* hit = nullptr;
        Slot* victim = &G.slot[0];
        for (Slot& sl : G.slot) {
            if (sl.handle && sl.width == width && sl.height == height && sl.depth == depth && sl.up == up_scale) hit = &sl;
            if (!sl.handle || (victim->handle && sl.used < victim->used)) victim = &sl;
        }
        if (!hit) {
            if (!G.quiet) std::cout << "Initialize ..." << std::endl;
            mif_sift_destroy(victim->handle);
            *victim = Slot();
            G.handle = nullptr;
            const int rc = mif_sift_create(G.device, width, height, depth, up_scale ? 1 : 0, &victim->handle);
            if (rc != MIF_OK) {
                std::cerr << "SIFT: " << mif_last_error() << std::endl;
                return rc;
            }
            victim->width = width; victim->height = height; victim->depth = depth; victim->up = up_scale;
            mif_sift_set_timing(victim->handle, 1);
            hit = victim;
        }
        hit->used = ++G.clock;
        G.handle = hit->handle;
        const size_t cap = (size_t)mif_sift_capacity(G.handle);
        if (G.kp.size() < cap * 4) {
            G.kp.resize(cap * 4);
            G.ori.resize(cap);
            G.desc.resize(cap * MIF_SIFT_DESC_DIM);
        }
        if (!G.quiet) std::cout << "Execute ..." << std::endl;
        return MIF_OK;
    }

    int run(size_t& elapsed_time, float offload, const unsigned char* image, int width, int height, int depth,
            std::vector<SIFT::sfeature>& features, std::vector<float>* descriptors, std::vector<float>* orientations,
            bool up_scale)
    {
        if (!offload_ok(offload)) return MIF_ERR_INVALID_ARGUMENT;
        int rc = init(image, width, height, depth, up_scale);
        if (rc != MIF_OK) return rc;
        Global& G = g();
        int counts[MIF_SIFT_OCTAVES] = {0, 0, 0};
        float ms = 0.0f;
        rc = mif_sift_extract(G.handle, image, G.kp.data(), descriptors ? G.ori.data() : nullptr,
                              descriptors ? G.desc.data() : nullptr, mif_sift_capacity(G.handle), counts, &ms);
        elapsed_time = (size_t)ms;
        {
            float st[MIF_SIFT_OCTAVES][MIF_SIFT_STAGES];
            if (G.history.size() < (size_t)1 << 22 && mif_sift_stage_times(G.handle, st) == MIF_OK)
                G.history.insert(G.history.end(), &st[0][0], &st[0][0] + MIF_SIFT_OCTAVES * MIF_SIFT_STAGES);
        }
        if (rc != MIF_OK) return rc;
        // octave-major order, SIFT/sift.cpp:1889-1911
        size_t row = 0;
        for (int o = 0; o < MIF_SIFT_OCTAVES; o++)
            for (int i = 0; i < counts[o]; i++, row++) {
                SIFT::sfeature f{};
                f.x = G.kp[4 * row + 0];
                f.y = G.kp[4 * row + 1];
                f.sigma = G.kp[4 * row + 2];
                f.scale_id = (size_t)G.kp[4 * row + 3];
                f.octave_id = (size_t)o;
                features.push_back(f);
            }
        if (descriptors) {
            descriptors->insert(descriptors->end(), G.desc.begin(), G.desc.begin() + row * MIF_SIFT_DESC_DIM);
            orientations->insert(orientations->end(), G.ori.begin(), G.ori.begin() + row);
        }
        if (!G.quiet) {
            std::cout << "--------------------------------------------------------" << std::endl;
            std::cout << "Elapsed_time_ms," << ms << std::endl;
            std::cout << "--------------------------------------------------------" << std::endl;
        }
        return MIF_OK;
    }

    // which: 0 = pyramid level sid, 1 = DoG image sid, 2 = input of the octave's scale space
    const float* probe(int which, float offload, const unsigned char* image, int width, int height, int depth, int oid,
                       int sid)
    {
        std::vector<SIFT::sfeature> feats;
        size_t ms = 0;
        if (run(ms, offload, image, width, height, depth, feats, nullptr, nullptr, false) != MIF_OK) return nullptr;
        Global& G = g();
        int w = 0, h = 0;
        if (mif_sift_octave_info(G.handle, oid, &w, &h, nullptr) != MIF_OK) return nullptr;
        G.probe.assign((size_t)w * h, 0.0f);
        const int rc = which == 1   ? mif_sift_copy_dog(G.handle, 0, oid, sid, G.probe.data())
                       : which == 2 ? mif_sift_copy_base(G.handle, 0, oid, G.probe.data())
                                    : mif_sift_copy_level(G.handle, 0, oid, sid, G.probe.data());
        return rc == MIF_OK ? G.probe.data() : nullptr;
    }
}

namespace SIFT
{
    int extract_features(size_t& elapsed_time, float offload, const unsigned char* input_color_image, int width,
                         int height, int color_depth, std::vector<sfeature>& features, bool up_scale)
    {
        return run(elapsed_time, offload, input_color_image, width, height, color_depth, features, nullptr, nullptr,
                   up_scale);
    }

    int extract_features(size_t& elapsed_time, float offload, const unsigned char* input_color_image, int width,
                         int height, int color_depth, std::vector<sfeature>& features,
                         std::vector<float>& descriptors, std::vector<float>& orientations, bool up_scale)
    {
        return run(elapsed_time, offload, input_color_image, width, height, color_depth, features, &descriptors,
                   &orientations, up_scale);
    }

    // SIFT/sift.cpp:915-957, 1674-1703: the reference writes, per task and octave, the files
    // <path><task><octave><scale>_off.dat / _cpu.dat / _gpu.dat with one line per call (iteration, value).
    // Here a stage of an octave takes the place of a task (the kernels are fused differently: see the
    // names below), the offload is always 1 and the CPU time 0; sift_stage_times.csv holds the last call.
    void store_task_exec_times(const std::string& path)
    {
        Global& G = g();
        if (G.handle == nullptr) return;
        float ms[MIF_SIFT_OCTAVES][MIF_SIFT_STAGES];
        if (mif_sift_stage_times(G.handle, ms) != MIF_OK) return;
        std::ofstream ofs(path + "sift_stage_times.csv");
        if (!ofs.is_open()) {
            std::cerr << "Couldn't open:\t" << path << "sift_stage_times.csv" << std::endl;
            return;
        }
        static const char* names[MIF_SIFT_STAGES] = {"base_image", "scale_space", "detector", "orientation_descriptor"};
        ofs << "octave,stage,gpu_ms" << std::endl;
        for (int o = 0; o < MIF_SIFT_OCTAVES; o++)
            for (int s = 0; s < MIF_SIFT_STAGES; s++) ofs << o << "," << names[s] << "," << ms[o][s] << std::endl;
        // base_image: kColor_interleaved_ch / kDown_pyramid / kResize_NN; scale_space: 6 x (kBlurH, kBlurV);
        // detector: kReset, kDiff, 3 x kDetector; orientation_descriptor: kOrientation, kDescriptor, kNormalize
        const size_t per_call = (size_t)MIF_SIFT_OCTAVES * MIF_SIFT_STAGES;
        const size_t calls = G.history.size() / per_call;
        for (int o = 0; o < MIF_SIFT_OCTAVES; o++)
            for (int s = 0; s < MIF_SIFT_STAGES; s++) {
                const std::string stem = path + names[s] + std::to_string(o) + "0";
                std::ofstream off(stem + "_off.dat"), cpu(stem + "_cpu.dat"), gpu(stem + "_gpu.dat");
                off << "iteration\toffload\n";
                cpu << "iteration\tcpu_time\n";
                gpu << "iteration\tgpu_time\n";
                for (size_t it = 0; it < calls; it++) {
                    off << it + 1 << "\t" << 1 << "\n";
                    cpu << it + 1 << "\t" << 0 << "\n";
                    gpu << it + 1 << "\t" << G.history[it * per_call + (size_t)o * MIF_SIFT_STAGES + s] << "\n";
                }
            }
    }

    size_t cnt_octaves() { return MIF_SIFT_OCTAVES; }
    size_t cnt_scales() { return MIF_SIFT_LEVELS; }

    const float* get_blur(float offload, const unsigned char* image, int width, int height, int color_depth, int oid,
                          int sid)
    {
        return probe(0, offload, image, width, height, color_depth, oid, sid);
    }

    const float* get_diff_blur(float offload, const unsigned char* image, int width, int height, int color_depth,
                               int oid, int sid)
    {
        return probe(1, offload, image, width, height, color_depth, oid, sid);
    }

    const float* get_gray(float offload, const unsigned char* image, int width, int height, int color_depth, int oid)
    {
        return probe(2, offload, image, width, height, color_depth, oid, 0);
    }

    std::vector<std::uint8_t> test_kResize(float offload, const unsigned char* image, int width, int height,
                                           int color_depth, float ratio_x, float ratio_y)
    {
        (void)offload;
        const int wd = (int)((float)width * ratio_x), hd = (int)((float)height * ratio_y);
        if (!image || wd < 1 || hd < 1 || color_depth < 1) return {};
        std::vector<std::uint8_t> out((size_t)wd * hd * color_depth);
        if (mif_probe_resize_u8(g().device, image, width, height, color_depth, ratio_x, ratio_y, out.data()) != MIF_OK)
            return {};
        return out;
    }

    std::vector<float> test_ColorToGray_Resize(float offload, const unsigned char* image, int width, int height,
                                               int color_depth, float ratio_x, float ratio_y)
    {
        (void)offload;
        const int wd = (int)((float)width * ratio_x), hd = (int)((float)height * ratio_y);
        if (!image || wd < 1 || hd < 1) return {};
        std::vector<float> out((size_t)wd * hd);
        if (mif_probe_gray_resize(g().device, image, width, height, color_depth, ratio_x, ratio_y, out.data()) != MIF_OK)
            return {};
        return out;
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
