// demo_coopcl_sift: the command line of the reference's SIFT demo (SIFT/main.cpp) on the GPU path.
//   -m object.pgm scene.pgm -f x   extract both, match, homography; report to stdout (+ -o report.json);
//                                  -r N repeats the whole sequence N more times without output and adds
//                                  the mean host-to-host milliseconds of every step to the report
//   -v frames -f x [-b N] [-g G]   per-frame extraction and matching against the previous frame;
//                                  frames = "file.raw:WxHxC" or a printf pattern of PGM/PPM files;
//                                  -b N sends batches of N frames through the GPU-resident pipeline;
//                                  -g G cuts the sequence into G blocks, one GPU and host thread each
//   -i image -f x                  extract one image, print the feature count
// -c (camera) needs a capture device and a display and is not available in this headless build.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <thread>
#include <vector>

#include "kpmatch.h"
#include "../../include/mi_features.h"
#include "parse_args.h"
#include "sift.h"

static const bool up_scale = false;

static std::string option(int argc, char** argv, const std::string& name)
{
    for (int i = 0; i + 1 < argc; i++)
        if (name == argv[i]) return argv[i + 1];
    return {};
}

// SIFT/main.cpp:176-206
static int call_feature_match_demo(const std::string& object_path, const std::string& scene_path, float offload,
                                   const std::string& report_path, int repeats)
{
    auto frame_object = mif::imread(object_path, true);
    auto frame_scene = mif::imread(scene_path, true);
    if (frame_object.empty() || frame_scene.empty()) {
        std::cerr << "Load image failed, fixme!" << std::endl;
        return -1;
    }
    std::vector<SIFT::sfeature> feats_object, feats_scene;
    std::vector<float> descs_object, descs_scene, orients_object, orients_scene;
    size_t elapsed_time = 0;
    int err = SIFT::extract_features(elapsed_time, offload, frame_object.data, frame_object.cols, frame_object.rows,
                                     frame_object.channels(), feats_object, descs_object, orients_object, up_scale);
    if (err != 0) return err;
    std::cout << "feats_object:\t" << feats_object.size() << std::endl;
    elapsed_time = 0;
    err = SIFT::extract_features(elapsed_time, offload, frame_scene.data, frame_scene.cols, frame_scene.rows,
                                 frame_scene.channels(), feats_scene, descs_scene, orients_scene, up_scale);
    if (err != 0) return err;
    std::cout << "feats_scene:\t" << feats_scene.size() << std::endl;

    std::vector<fmatches::sift_feat> kp_object, kp_scene;
    fmatches::ConvertToSift_feat<SIFT::sfeature>(feats_object, orients_object, descs_object, kp_object);
    fmatches::ConvertToSift_feat<SIFT::sfeature>(feats_scene, orients_scene, descs_scene, kp_scene);
    const auto report = fmatches::Plot_Img_Matches_ocv(frame_object, frame_scene, kp_object, kp_scene, up_scale);
    if (report.has_homography) {
        std::cout << "Homography inliers:\t" << report.inliers << std::endl;
        for (int i = 0; i < 4; i++)
            std::cout << "corner" << i << ":\t" << report.scene_corners[i].x << "\t" << report.scene_corners[i].y
                      << std::endl;
    }
    // -r N: the same sequence N more times (pipelines and matcher workspace exist by now), timed step by
    // step with the host's clock: pixels in host memory -> features, matches, homography in host memory
    fmatches::Timing tm;
    if (repeats > 0) {
        using clock = std::chrono::steady_clock;
        auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        SIFT::set_quiet(true);
        fmatches::set_quiet(true);
        tm.total_min = 1e30;
        for (int r = 0; r < repeats; r++) {
            std::vector<SIFT::sfeature> fo, fs;
            std::vector<float> dso, dss, oo, os;
            const auto t0 = clock::now();
            err = SIFT::extract_features(elapsed_time, offload, frame_object.data, frame_object.cols, frame_object.rows,
                                         frame_object.channels(), fo, dso, oo, up_scale);
            const auto t1 = clock::now();
            if (err == 0)
                err = SIFT::extract_features(elapsed_time, offload, frame_scene.data, frame_scene.cols, frame_scene.rows,
                                             frame_scene.channels(), fs, dss, os, up_scale);
            const auto t2 = clock::now();
            if (err != 0) return err;
            std::vector<fmatches::sift_feat> ko, ks;
            fmatches::ConvertToSift_feat<SIFT::sfeature>(fo, oo, dso, ko);
            fmatches::ConvertToSift_feat<SIFT::sfeature>(fs, os, dss, ks);
            const auto t3 = clock::now();
            const auto rep = fmatches::Plot_Img_Matches_ocv(frame_object, frame_scene, ko, ks, up_scale);
            const auto t4 = clock::now();
            if (rep.good_matches.size() != report.good_matches.size() || fo.size() != feats_object.size()) {
                std::cerr << "repeat " << r << " differs from the first run" << std::endl;
                return -1;
            }
            tm.extract_object += ms(t0, t1);
            tm.extract_scene += ms(t1, t2);
            tm.convert += ms(t2, t3);
            tm.match += rep.match_ms;
            tm.homography += rep.homography_ms;
            tm.total += ms(t0, t4);
            tm.total_min = std::min(tm.total_min, ms(t0, t4));
        }
        SIFT::set_quiet(false);
        fmatches::set_quiet(false);
        tm.repeats = repeats;
        for (double* v : {&tm.extract_object, &tm.extract_scene, &tm.convert, &tm.match, &tm.homography, &tm.total})
            *v /= repeats;
        tm.feats_object = feats_object.size();
        tm.feats_scene = feats_scene.size();
        std::cout << "Repeats:\t" << repeats << "\tmean total_ms:\t" << tm.total << "\textract:\t"
                  << tm.extract_object + tm.extract_scene << "\tmatch:\t" << tm.match << "\thomography:\t"
                  << tm.homography << std::endl;
    }
    if (!report_path.empty()) fmatches::write_report_json(report_path, report, tm);
    return 0;
}

// SIFT/main.cpp:103-153; each frame is also matched against the previous one
static int call_stream_demo(const std::string& input, float offload)
{
    mif::FrameReader reader(input);
    if (!reader.isOpened()) {
        std::cerr << "Error opening video stream or file:\t" << input << std::endl;
        return -1;
    }
    size_t acc_elapsed_time = 0, cnt_frames = 0, total_good = 0;
    const size_t max_frames = 100;
    std::vector<fmatches::sift_feat> prev;
    mif::Image prev_frame;
    const auto t0 = std::chrono::steady_clock::now();
    while (cnt_frames < max_frames) {
        mif::Image frame;
        if (!reader.read(frame)) break;
        std::vector<SIFT::sfeature> feats;
        std::vector<float> descs, orients;
        size_t elapsed_time = 0;
        const int err = SIFT::extract_features(elapsed_time, offload, frame.data, frame.cols, frame.rows,
                                               frame.channels(), feats, descs, orients, up_scale);
        acc_elapsed_time += elapsed_time;
        std::vector<fmatches::sift_feat> cur;
        if (err == 0) fmatches::ConvertToSift_feat<SIFT::sfeature>(feats, orients, descs, cur);
        std::cout << "Frame id<" << cnt_frames << "> Features: <" << cur.size() << ">" << std::endl;
        if (!prev.empty() && !cur.empty())
            total_good += fmatches::Plot_Img_Matches_ocv(frame, prev_frame, cur, prev, up_scale).good_matches.size();
        prev.swap(cur);
        prev_frame = frame;
        cnt_frames++;
    }
    if (cnt_frames == 0) return -1;
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "Average time_ms:\t" << acc_elapsed_time / cnt_frames << std::endl;
    std::cout << "Overall time_ms:\t" << acc_elapsed_time << std::endl;
    std::cout << "Frames:\t" << cnt_frames << "\tgood matches:\t" << total_good << "\twall_ms:\t" << wall_ms << std::endl;
    SIFT::store_task_exec_times("./");
    return 0;
}

// Frames [start, hi) of the sequence through one frame-sequence pipeline (mif_video_*) on `device`, in
// batches of `batch`: the frames stay on the GPU, the upload of one batch overlaps the work on the one
// before, and every frame is matched against its predecessor there. One record per frame in `out`.
static int run_block(const std::string& input, int device, int batch, size_t start, size_t hi,
                     std::vector<mif_video_frame>& out, std::string& error)
{
    mif::FrameReader reader(input);
    if (!reader.isOpened()) { error = "cannot open " + input; return -1; }
    reader.seek(start);
    mif_video* pipe = nullptr;
    int rows = 0, cols = 0, channels = 0;
    // a submitted buffer must stay untouched until the call after next has returned: three in rotation
    std::vector<unsigned char> staging[3];
    size_t at = start, submitted = 0;
    bool more = true;
    auto fail = [&](const char* what) {
        error = std::string(what) + ": " + mif_last_error();
        if (pipe) mif_video_destroy(pipe);
        return -1;
    };
    while (more && at < hi) {
        std::vector<unsigned char>& buf = staging[submitted % 3];
        int n = 0;
        while (n < batch && at < hi) {
            mif::Image frame;
            if (!reader.read(frame)) { more = false; break; }
            if (!pipe) {
                rows = frame.rows; cols = frame.cols; channels = frame.channels();
                if (mif_video_create(device, cols, rows, channels, batch, &pipe) != MIF_OK) return fail("pipeline");
            }
            if (frame.rows != rows || frame.cols != cols || frame.channels() != channels) {
                error = "frame " + std::to_string(at) + " differs in size from the first one";
                more = false;
                break;
            }
            const size_t bytes = (size_t)rows * cols * channels;
            if (buf.size() < bytes * batch) buf.resize(bytes * batch);
            std::copy(frame.data, frame.data + bytes, buf.begin() + bytes * n);
            n++;
            at++;
        }
        if (n == 0) break;
        if (mif_video_submit_host(pipe, buf.data(), n) != MIF_OK) return fail("submit");
        submitted++;
    }
    if (at == start || !pipe) { if (error.empty()) error = "no frames"; return -1; }
    out.resize(at - start);
    int got = 0;
    if (mif_video_collect(pipe, out.data(), (int)out.size(), &got) != MIF_OK) return fail("collect");
    out.resize((size_t)got);
    mif_video_destroy(pipe);
    return 0;
}

// The loop of SIFT/main.cpp:103-153 for whole batches of frames (-b N), optionally sharded over GPUs
// (-g G): the sequence is cut into G contiguous blocks, block g runs on GPU g (modulo the GPUs present)
// in a host thread of its own with a pipeline of its own, and nothing is exchanged between them. A block
// that does not start the sequence is extended by the frame before it, so that its first frame is matched
// against the same predecessor as in a one-GPU run; that extra frame's record is dropped. Prints the
// per-frame lines of SIFT/main.cpp:129 once the results are in.
static int call_stream_demo_batched(const std::string& input, int batch, int gpus)
{
    mif::FrameReader reader(input);
    if (!reader.isOpened()) {
        std::cerr << "Error opening video stream or file:\t" << input << std::endl;
        return -1;
    }
    const size_t max_frames = 100;
    const size_t total = reader.count(max_frames);
    if (total == 0) return -1;
    const int ndev = mif_device_count();
    if (ndev < 1) { std::cerr << "no GPU" << std::endl; return -1; }
    const size_t G = (size_t)std::max(1, std::min<int>(gpus, (int)total));
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::vector<mif_video_frame>> part(G);
    std::vector<std::string> errors(G);
    std::vector<int> codes(G, 0);
    std::vector<size_t> first(G + 1, 0);
    for (size_t g = 0; g < G; g++) first[g + 1] = first[g] + total / G + (g < total % G ? 1 : 0);
    auto work = [&](size_t g) {
        const size_t lo = first[g], hi = first[g + 1];
        const size_t start = lo > 0 ? lo - 1 : lo;
        codes[g] = run_block(input, (int)(g % (size_t)ndev), batch, start, hi, part[g], errors[g]);
        if (codes[g] == 0 && start < lo) part[g].erase(part[g].begin());
    };
    if (G == 1) {
        work(0);
    } else {
        std::vector<std::thread> threads;
        for (size_t g = 0; g < G; g++) threads.emplace_back(work, g);
        for (auto& t : threads) t.join();
    }
    for (size_t g = 0; g < G; g++) {
        if (codes[g] != 0) {
            std::cerr << "block " << g << ":\t" << errors[g] << std::endl;
            return -1;
        }
    }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    size_t total_good = 0, got = 0;
    for (size_t g = 0; g < G; g++) {
        for (const mif_video_frame& f : part[g]) {
            const int feats = f.status == 0 ? f.keypoints[0] + f.keypoints[1] + f.keypoints[2] : 0;
            std::cout << "Frame id<" << got << "> Features: <" << feats << ">" << std::endl;
            if (f.good_matches > 0) total_good += (size_t)f.good_matches;
            got++;
        }
    }
    std::cout << "Average time_ms:\t" << wall_ms / got << std::endl;
    std::cout << "Overall time_ms:\t" << wall_ms << std::endl;
    std::cout << "Frames:\t" << got << "\tgood matches:\t" << total_good << "\twall_ms:\t" << wall_ms
              << "\tGPUs:\t" << std::min<size_t>(G, (size_t)ndev) << "\tblocks:\t" << G << std::endl;
    return 0;
}

// SIFT/main.cpp:155-174
static int call_single_img_demo(const std::string& path, float offload)
{
    auto frame = mif::imread(path, false);
    if (frame.empty()) {
        std::cerr << "Load image failed, fixme!" << std::endl;
        return -1;
    }
    std::vector<SIFT::sfeature> feats;
    std::vector<float> descs, orients;
    size_t elapsed_time = 0;
    const int err = SIFT::extract_features(elapsed_time, offload, frame.data, frame.cols, frame.rows, frame.channels(),
                                           feats, descs, orients, up_scale);
    if (err != 0) return err;
    std::cout << "Features: <" << feats.size() << ">" << std::endl;
    return 0;
}

int main(int argc, char** argv)
{
    iargs::app_args args;
    if (!args.parse(argc, argv)) return iargs::app_args::usage(std::cerr);
    const std::string report = option(argc, argv, "-o");
    if (args.app_mode == "-h")
        iargs::app_args::usage(std::cout);
    else if (args.app_mode == "-i")
        return call_single_img_demo(args.img_path_a, args.offload);
    else if (args.app_mode == "-v") {
        const std::string b = option(argc, argv, "-b");
        const std::string g = option(argc, argv, "-g");
        const int gpus = g.empty() ? 1 : std::atoi(g.c_str());
        if (gpus > 1 && b.empty()) return call_stream_demo_batched(args.img_path_a, 16, gpus);
        if (!b.empty() && std::atoi(b.c_str()) > 0)
            return call_stream_demo_batched(args.img_path_a, std::atoi(b.c_str()), std::max(1, gpus));
        return call_stream_demo(args.img_path_a, args.offload);
    }
    else if (args.app_mode == "-c") {
        std::cerr << "Error opening video stream or file:\t(camera input is not available in this build)" << std::endl;
        return -1;
    } else if (args.app_mode == "-m")
        return call_feature_match_demo(args.img_path_a, args.img_path_b, args.offload, report,
                                       std::max(0, std::atoi(option(argc, argv, "-r").c_str())));
    return 0;
}
