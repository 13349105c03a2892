// demo_coopcl_Surf: the command line of the reference's SURF demo (SURF/coopcl_surf/main.cpp) on the GPU path.
//   -m object.pgm scene.pgm -f x   extract both, ratio-test matching, homography (+ -o report.json);
//                                  -r N repeats the whole sequence N more times without output and adds
//                                  the mean host-to-host milliseconds of every step to the report
//   -v frames -f x                 per-frame extraction; frames = "file.raw:WxHxC" or a printf pattern
//   -i image -f x                  extract one image, print the point count
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "ccl_surf.h"
#include "kpmatch.h"
#include "parse_args.h"

static std::string option(int argc, char** argv, const std::string& name)
{
    for (int i = 0; i + 1 < argc; i++)
        if (name == argv[i]) return argv[i + 1];
    return {};
}

// SURF/coopcl_surf/main.cpp:184-241: the reference builds its OpenCL tasks from a list of
// (file, kernel name) pairs; the names are kept so that the call is the same
static bool initialize_surf_driver(float offload)
{
    const std::vector<std::pair<std::string, std::string>> kernels = {
        {"createDescriptors_kernel.cl", "createDescriptors"}, {"getOrientation_kernels.cl", "getOrientationStep1"},
        {"getOrientation_kernels.cl", "getOrientationStep2"}, {"hessianDet_kernel.cl", "hessian_det"},
        {"integralImage_kernels.cl", "scan"}, {"integralImage_kernels.cl", "transpose"},
        {"nonMaxSuppression_kernel.cl", "non_max_supression"}, {"normalizeDescriptors_kernel.cl", "normalizeDescriptors"}};
    return ccl_surf::initialize_surf_driver(kernels, false, offload);
}

static void copy_ipoints(const ccl_surf::IpVec& src, std::vector<fmatches::Ipoint>& dst)
{
    for (const auto& item : src) {
        fmatches::Ipoint p;
        std::memcpy(&p, &item, sizeof(p));
        dst.push_back(p);
    }
}

// SURF/coopcl_surf/main.cpp:141-182
static int call_feature_match_demo(const std::string& object_path, const std::string& scene_path,
                                   const std::string& report_path, int repeats)
{
    auto frame_object = mif::imread(object_path, true);
    auto frame_scene = mif::imread(scene_path, true);
    if (frame_object.empty() || frame_scene.empty()) {
        std::cerr << "Load image failed, fixme!" << std::endl;
        return -1;
    }
    std::vector<fmatches::Ipoint> feats_object, feats_scene;
    copy_ipoints(ccl_surf::run(frame_object), feats_object);
    std::cout << "feats_object:\t" << feats_object.size() << std::endl;
    copy_ipoints(ccl_surf::run(frame_scene), feats_scene);
    std::cout << "feats_scene:\t" << feats_scene.size() << std::endl;
    const auto report = fmatches::Plot_Img_Matches_ocv(frame_object, frame_scene, feats_object, feats_scene);
    if (report.has_homography) {
        std::cout << "Homography inliers:\t" << report.inliers << std::endl;
        for (int i = 0; i < 4; i++)
            std::cout << "corner" << i << ":\t" << report.scene_corners[i].x << "\t" << report.scene_corners[i].y
                      << std::endl;
    }
    // -r N: the same sequence N more times (pipelines and matcher workspace exist by now), timed step by
    // step with the host's clock: pixels in host memory -> points, matches, homography in host memory
    fmatches::Timing tm;
    if (repeats > 0) {
        using clock = std::chrono::steady_clock;
        auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        ccl_surf::set_quiet(true);
        fmatches::set_quiet(true);
        tm.total_min = 1e30;
        for (int r = 0; r < repeats; r++) {
            const auto t0 = clock::now();
            const auto io = ccl_surf::run(frame_object);
            const auto t1 = clock::now();
            const auto is = ccl_surf::run(frame_scene);
            const auto t2 = clock::now();
            std::vector<fmatches::Ipoint> fo, fs;
            copy_ipoints(io, fo);
            copy_ipoints(is, fs);
            const auto t3 = clock::now();
            const auto rep = fmatches::Plot_Img_Matches_ocv(frame_object, frame_scene, fo, fs);
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
        ccl_surf::set_quiet(false);
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

// SURF/coopcl_surf/main.cpp:77-139
static int call_stream_demo(const std::string& input)
{
    mif::FrameReader reader(input);
    if (!reader.isOpened()) {
        std::cerr << "Error opening video stream or file:\t" << input << std::endl;
        return -1;
    }
    size_t acc_elapsed_time = 0, cnt_frames = 0;
    const size_t max_frames = 100;
    while (cnt_frames < max_frames) {
        mif::Image frame;
        if (!reader.read(frame)) break;
        const auto start = std::chrono::system_clock::now();
        const auto ipts = ccl_surf::run(frame);
        const auto end = std::chrono::system_clock::now();
        acc_elapsed_time += (size_t)std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count();
        std::cout << "Frame id<" << cnt_frames << "> Features: <" << ipts.size() << ">" << std::endl;
        cnt_frames++;
    }
    if (cnt_frames == 0) return -1;
    std::cout << "Average elapsed_time:\t" << acc_elapsed_time / cnt_frames << " ms" << std::endl;
    std::cout << "Overall time_ms:\t" << acc_elapsed_time << std::endl;
    return 0;
}

static int call_single_img_demo(const std::string& path)
{
    auto frame = mif::imread(path, false);
    if (frame.empty()) {
        std::cerr << "Load image failed, fixme!" << std::endl;
        return -1;
    }
    std::cout << "Features: <" << ccl_surf::run(frame).size() << ">" << std::endl;
    return 0;
}

int main(int argc, char** argv)
{
    iargs::app_args args;
    if (!args.parse(argc, argv)) return iargs::app_args::usage(std::cerr);
    if (args.app_mode == "-h") {
        iargs::app_args::usage(std::cout);
        return 0;
    }
    if (!initialize_surf_driver(args.offload)) return -1;
    const std::string report = option(argc, argv, "-o");
    if (args.app_mode == "-i") return call_single_img_demo(args.img_path_a);
    if (args.app_mode == "-v") return call_stream_demo(args.img_path_a);
    if (args.app_mode == "-c") {
        std::cerr << "Error opening video stream or file:\t(camera input is not available in this build)" << std::endl;
        return -1;
    }
    if (args.app_mode == "-m")
        return call_feature_match_demo(args.img_path_a, args.img_path_b, report,
                                       std::max(0, std::atoi(option(argc, argv, "-r").c_str())));
    return 0;
}
