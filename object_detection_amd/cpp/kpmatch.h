// Feature conversion, matching, good-match filters and homography: the host
// side of the reference's common/kpmatch.h without OpenCV. The FLANN calls are
// replaced by the exact GPU matcher (mif_match_top2), cv::findHomography by a
// seeded RANSAC + normalised DLT, and the imshow / drawMatches calls by a text
// report; names and the data that flows between the steps are the reference's.
#pragma once
#include <string>
#include <vector>

#include "image_view.h"

namespace fmatches
{
    // common/kpmatch.h:7-16. `descriptor` starts with ONE zero element, as in the reference
    // (std::vector<float> descriptor{ 0 }); ConvertToSift_feat appends the 128 values behind it.
    struct sift_feat
    {
        float x{ 0 };
        float y{ 0 };
        float sigma{ 0 };
        float orient{ 0 };
        float octave_id{ 0 };
        float scale{ 0 };
        std::vector<float> descriptor{ 0 };
    };

    // common/kpmatch.h:19-27
    typedef struct {
        float x;
        float y;
        float scale;
        float orientation;
        int laplacian;
        int clusterIndex;
        float descriptor[64];
    } Ipoint;

    // cv::DMatch
    struct DMatch
    {
        int queryIdx;
        int trainIdx;
        float distance;
    };

    struct Point2f { float x, y; };

    // common/kpmatch.h:108-131
    template<typename T>
    void ConvertToSift_feat(const std::vector<T>& feats, const std::vector<float>& orientations,
                            const std::vector<float>& descriptors, std::vector<sift_feat>& kp_out)
    {
        const auto cnt_kp = feats.size();
        if (cnt_kp == 0) return;
        const auto desc_size = descriptors.size() / cnt_kp;
        for (size_t i = 0; i < cnt_kp; i++) {
            sift_feat sf;
            sf.x = feats[i].x;
            sf.y = feats[i].y;
            sf.sigma = feats[i].sigma;
            sf.scale = (float)feats[i].scale_id;
            sf.octave_id = (float)feats[i].octave_id;
            sf.orient = orientations[i];
            for (size_t d = 0; d < desc_size; d++) sf.descriptor.push_back(descriptors[d + i * desc_size]);
            kp_out.push_back(sf);
        }
    }

    // The descriptor matrices ConvertToCVKeyPoint builds (common/kpmatch.h:133-159): row-major
    // N x 128 holding the FIRST 128 elements of sift_feat::descriptor (so a row is [0, d0 .. d126]),
    // resp. N x 64 for Ipoints; keypoint positions go to `pts`.
    void ConvertToKeyPoint(const std::vector<sift_feat>& kp_in, std::vector<Point2f>& pts, std::vector<float>& desc);
    void ConvertToKeyPoint(const std::vector<Ipoint>& kp_in, std::vector<Point2f>& pts, std::vector<float>& desc);

    struct MatchReport
    {
        std::vector<DMatch> good_matches;
        double min_dist = 100, max_dist = 0;
        bool has_homography = false;
        double H[9] = { 0 };                 // row-major 3x3, maps object to scene
        int inliers = 0;
        Point2f scene_corners[4] = {};       // object corners (0,0) (w,0) (w,h) (0,h) mapped into the scene
        // host wall-clock of the two steps of the call that made this report
        double match_ms = 0;                 // descriptor matrices, GPU search (host buffers in, results out), filter
        double homography_ms = 0;            // RANSAC + DLT + corner projection
    };

    // SIFT overload of Plot_Img_Matches_ocv (common/kpmatch.h:162-268) without the drawing: nearest
    // neighbour per object feature, keep dist <= max(2.5 * min_dist, 0.35), homography by RANSAC,
    // projected object corners. Prints the reference's "-- Max dist", "-- Min dist", "Good matches:" lines.
    MatchReport Plot_Img_Matches_ocv(const mif::Image& img_object, const mif::Image& img_scene,
                                     const std::vector<sift_feat>& kp_object, const std::vector<sift_feat>& kp_scene,
                                     const bool up_scale = false);

    // SURF overload (common/kpmatch.h:270-320): 2-NN, Lowe ratio 0.75, "Found stable matches:" line;
    // additionally estimates the homography the same way as the SIFT overload.
    MatchReport Plot_Img_Matches_ocv(const mif::Image& img_object, const mif::Image& img_scene,
                                     const std::vector<Ipoint>& kp_object, const std::vector<Ipoint>& kp_scene,
                                     const bool up_scale = false);

    // RANSAC homography (stand-in for cv::findHomography(obj, scene, cv::RANSAC), common/kpmatch.h:233):
    // 4-point samples drawn from std::mt19937(seed), normalised DLT, reprojection threshold 3 px, final
    // least-squares refit on the inliers. Returns false with fewer than 4 pairs or no consistent model.
    bool findHomographyRansac(const std::vector<Point2f>& obj, const std::vector<Point2f>& scene, double H[9],
                              std::vector<char>* inlier_mask = nullptr, unsigned seed = 12345);

    // cv::perspectiveTransform for one point
    Point2f perspectiveTransform(const Point2f& p, const double H[9]);

    // Writes a report as JSON (matches, distances, homography, corners).
    bool write_report_json(const std::string& path, const MatchReport& report);

    void set_device(int device);
    // true: the "-- Max dist" / "Good matches:" lines of the reference are not printed (timing loops)
    void set_quiet(bool quiet);

    // Wall-clock summary of a demo's repeat loop (-r N), written into the JSON report.
    struct Timing
    {
        int repeats = 0;
        // means over the repeats, host to host, in ms
        double extract_object = 0, extract_scene = 0, convert = 0, match = 0, homography = 0, total = 0;
        double total_min = 0;                // fastest repeat
        size_t feats_object = 0, feats_scene = 0;
    };
    bool write_report_json(const std::string& path, const MatchReport& report, const Timing& timing);
}
