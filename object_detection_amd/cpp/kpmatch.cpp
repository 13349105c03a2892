#include "kpmatch.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <limits>
#include <random>

#include "../../include/mi_features.h"

namespace fmatches
{
    namespace
    {
        int g_device = 0;
        bool g_quiet = false;

        // The matcher workspace is kept from call to call and grows with the largest request seen
        // (the reference builds a cv::FlannBasedMatcher per call, common/kpmatch.h:174).
        struct MatcherCache
        {
            mif_matcher* m = nullptr;
            int device = -1, max_q = 0, max_d = 0;
            ~MatcherCache() { mif_match_destroy(m); }
        };

        double ms_since(const std::chrono::steady_clock::time_point& t0)
        {
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }

        // exact 2-NN of every query row on the GPU
        bool gpu_top2(const std::vector<float>& q, int nq, const std::vector<float>& db, int nd, int dim,
                      std::vector<int>& idx2, std::vector<float>& dist2)
        {
            static MatcherCache cache;
            idx2.assign((size_t)nq * 2, -1);
            dist2.assign((size_t)nq * 2, std::numeric_limits<float>::infinity());
            if (nq == 0) return true;
            if (!cache.m || cache.device != g_device || cache.max_q < nq || cache.max_d < nd) {
                mif_match_destroy(cache.m);
                cache.m = nullptr;
                cache.max_q = std::max(cache.max_q, std::max(nq, 1));
                cache.max_d = std::max(cache.max_d, std::max(nd, 1));
                cache.device = g_device;
                if (mif_match_create(g_device, cache.max_q, cache.max_d, &cache.m) != MIF_OK) {
                    std::cerr << "matcher: " << mif_last_error() << std::endl;
                    cache.max_q = cache.max_d = 0;
                    return false;
                }
            }
            const int rc = mif_match_top2(cache.m, q.data(), nq, db.data(), nd, dim, dim, idx2.data(), dist2.data());
            if (rc != MIF_OK) std::cerr << "matcher: " << mif_last_error() << std::endl;
            return rc == MIF_OK;
        }

        // ---- small dense linear algebra for the DLT --------------------------------------------

        // eigenvector of the smallest eigenvalue of a symmetric 9x9 matrix (cyclic Jacobi)
        void smallest_eigenvector9(double A[9][9], double v_out[9])
        {
            double V[9][9];
            for (int i = 0; i < 9; i++)
                for (int j = 0; j < 9; j++) V[i][j] = i == j ? 1.0 : 0.0;
            for (int sweep = 0; sweep < 60; sweep++) {
                double off = 0.0;
                for (int p = 0; p < 9; p++)
                    for (int q = p + 1; q < 9; q++) off += A[p][q] * A[p][q];
                if (off < 1e-30) break;
                for (int p = 0; p < 9; p++)
                    for (int q = p + 1; q < 9; q++) {
                        if (std::fabs(A[p][q]) < 1e-300) continue;
                        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                        for (int k = 0; k < 9; k++) {
                            const double akp = A[k][p], akq = A[k][q];
                            A[k][p] = c * akp - s * akq;
                            A[k][q] = s * akp + c * akq;
                        }
                        for (int k = 0; k < 9; k++) {
                            const double apk = A[p][k], aqk = A[q][k];
                            A[p][k] = c * apk - s * aqk;
                            A[q][k] = s * apk + c * aqk;
                        }
                        for (int k = 0; k < 9; k++) {
                            const double vkp = V[k][p], vkq = V[k][q];
                            V[k][p] = c * vkp - s * vkq;
                            V[k][q] = s * vkp + c * vkq;
                        }
                    }
            }
            int best = 0;
            for (int i = 1; i < 9; i++)
                if (A[i][i] < A[best][best]) best = i;
            for (int k = 0; k < 9; k++) v_out[k] = V[k][best];
        }

        struct Norm { double cx, cy, s; };

        Norm normalisation(const std::vector<Point2f>& p, const std::vector<int>& ids)
        {
            Norm n{0, 0, 1};
            for (int i : ids) { n.cx += p[i].x; n.cy += p[i].y; }
            n.cx /= ids.size(); n.cy /= ids.size();
            double d = 0;
            for (int i : ids) d += std::hypot(p[i].x - n.cx, p[i].y - n.cy);
            d /= ids.size();
            n.s = d > 1e-12 ? std::sqrt(2.0) / d : 1.0;
            return n;
        }

        // normalised DLT over the pairs `ids` (Hartley); false when degenerate
        bool dlt(const std::vector<Point2f>& a, const std::vector<Point2f>& b, const std::vector<int>& ids, double H[9])
        {
            if (ids.size() < 4) return false;
            const Norm na = normalisation(a, ids), nb = normalisation(b, ids);
            double M[9][9] = {{0}};
            for (int i : ids) {
                const double x = (a[i].x - na.cx) * na.s, y = (a[i].y - na.cy) * na.s;
                const double u = (b[i].x - nb.cx) * nb.s, v = (b[i].y - nb.cy) * nb.s;
                const double r1[9] = {-x, -y, -1, 0, 0, 0, u * x, u * y, u};
                const double r2[9] = {0, 0, 0, -x, -y, -1, v * x, v * y, v};
                for (int p = 0; p < 9; p++)
                    for (int q = 0; q < 9; q++) M[p][q] += r1[p] * r1[q] + r2[p] * r2[q];
            }
            double h[9];
            smallest_eigenvector9(M, h);
            // H = Tb^-1 * Hn * Ta
            const double Ta[9] = {na.s, 0, -na.s * na.cx, 0, na.s, -na.s * na.cy, 0, 0, 1};
            const double Tbi[9] = {1 / nb.s, 0, nb.cx, 0, 1 / nb.s, nb.cy, 0, 0, 1};
            double T1[9], T2[9];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    T1[3 * r + c] = 0;
                    for (int k = 0; k < 3; k++) T1[3 * r + c] += h[3 * r + k] * Ta[3 * k + c];
                }
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    T2[3 * r + c] = 0;
                    for (int k = 0; k < 3; k++) T2[3 * r + c] += Tbi[3 * r + k] * T1[3 * k + c];
                }
            if (std::fabs(T2[8]) < 1e-12) return false;
            for (int k = 0; k < 9; k++) H[k] = T2[k] / T2[8];
            for (int k = 0; k < 9; k++)
                if (!std::isfinite(H[k])) return false;
            return true;
        }

        double reproj_error(const Point2f& p, const Point2f& q, const double H[9])
        {
            const double w = H[6] * p.x + H[7] * p.y + H[8];
            if (std::fabs(w) < 1e-12) return std::numeric_limits<double>::infinity();
            const double x = (H[0] * p.x + H[1] * p.y + H[2]) / w, y = (H[3] * p.x + H[4] * p.y + H[5]) / w;
            return std::hypot(x - q.x, y - q.y);
        }

        void finish_report(MatchReport& rep, const mif::Image& img_object, const std::vector<Point2f>& pts_object,
                           const std::vector<Point2f>& pts_scene, bool up_scale)
        {
            // common/kpmatch.h:223-243
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<Point2f> obj, scene;
            for (const DMatch& m : rep.good_matches) {
                obj.push_back(pts_object[m.queryIdx]);
                scene.push_back(pts_scene[m.trainIdx]);
            }
            std::vector<char> mask;
            rep.has_homography = findHomographyRansac(obj, scene, rep.H, &mask);
            if (!rep.has_homography) {
                std::cerr << "Couldn't findHomography ... fixme!! " << std::endl;
                rep.homography_ms = ms_since(t0);
                return;
            }
            rep.inliers = (int)std::count(mask.begin(), mask.end(), (char)1);
            const float f = up_scale ? 2.0f : 1.0f;
            const float cols = img_object.cols * f, rows = img_object.rows * f;
            const Point2f corners[4] = {{0, 0}, {cols, 0}, {cols, rows}, {0, rows}};
            for (int i = 0; i < 4; i++) rep.scene_corners[i] = perspectiveTransform(corners[i], rep.H);
            rep.homography_ms = ms_since(t0);
        }
    }

    void set_device(int device) { g_device = device; }
    void set_quiet(bool quiet) { g_quiet = quiet; }

    void ConvertToKeyPoint(const std::vector<sift_feat>& kp_in, std::vector<Point2f>& pts, std::vector<float>& desc)
    {
        desc.assign(kp_in.size() * 128, 0.0f);
        pts.clear();
        size_t y = 0;
        for (const auto& item : kp_in) {
            pts.push_back(Point2f{item.x, item.y});
            for (size_t x = 0; x < 128 && x < item.descriptor.size(); x++) desc[y * 128 + x] = item.descriptor[x];
            y++;
        }
    }

    void ConvertToKeyPoint(const std::vector<Ipoint>& kp_in, std::vector<Point2f>& pts, std::vector<float>& desc)
    {
        desc.assign(kp_in.size() * 64, 0.0f);
        pts.clear();
        size_t y = 0;
        for (const auto& item : kp_in) {
            pts.push_back(Point2f{item.x, item.y});
            for (size_t x = 0; x < 64; x++) desc[y * 64 + x] = item.descriptor[x];
            y++;
        }
    }

    Point2f perspectiveTransform(const Point2f& p, const double H[9])
    {
        const double w = H[6] * p.x + H[7] * p.y + H[8];
        return Point2f{(float)((H[0] * p.x + H[1] * p.y + H[2]) / w), (float)((H[3] * p.x + H[4] * p.y + H[5]) / w)};
    }

    bool findHomographyRansac(const std::vector<Point2f>& obj, const std::vector<Point2f>& scene, double H[9],
                              std::vector<char>* inlier_mask, unsigned seed)
    {
        const int n = (int)obj.size();
        if (n < 4 || scene.size() != obj.size()) return false;
        const double thr = 3.0;
        std::mt19937 rng(seed);
        std::vector<int> best_inl;
        int max_iter = 2000;
        for (int it = 0; it < max_iter; it++) {
            std::vector<int> ids;
            while (ids.size() < 4) {
                const int c = (int)(rng() % (unsigned)n);
                if (std::find(ids.begin(), ids.end(), c) == ids.end()) ids.push_back(c);
            }
            double Ht[9];
            if (!dlt(obj, scene, ids, Ht)) continue;
            std::vector<int> inl;
            for (int i = 0; i < n; i++)
                if (reproj_error(obj[i], scene[i], Ht) < thr) inl.push_back(i);
            if (inl.size() > best_inl.size()) {
                best_inl.swap(inl);
                // adaptive stop at 99.5 % confidence
                const double w = (double)best_inl.size() / n;
                const double denom = std::log(std::max(1e-12, 1.0 - std::pow(w, 4)));
                if (denom < 0) max_iter = std::min(max_iter, (int)std::ceil(std::log(1 - 0.995) / denom) + it + 1);
            }
        }
        if (best_inl.size() < 4) return false;
        if (!dlt(obj, scene, best_inl, H)) return false;
        if (inlier_mask) {
            inlier_mask->assign(n, 0);
            for (int i = 0; i < n; i++)
                if (reproj_error(obj[i], scene[i], H) < thr) (*inlier_mask)[i] = 1;
        }
        return true;
    }

    MatchReport Plot_Img_Matches_ocv(const mif::Image& img_object, const mif::Image&,
                                     const std::vector<sift_feat>& kp_object, const std::vector<sift_feat>& kp_scene,
                                     const bool up_scale)
    {
        MatchReport rep;
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<Point2f> pts_object, pts_scene;
        std::vector<float> desc_object, desc_scene;
        ConvertToKeyPoint(kp_object, pts_object, desc_object);
        ConvertToKeyPoint(kp_scene, pts_scene, desc_scene);
        std::vector<int> idx2;
        std::vector<float> dist2;
        const int nq = (int)kp_object.size(), nd = (int)kp_scene.size();
        if (!gpu_top2(desc_object, nq, desc_scene, nd, 128, idx2, dist2)) return rep;
        // common/kpmatch.h:177-202
        for (int i = 0; i < nq; i++) {
            const double dist = dist2[2 * i];
            if (dist < rep.min_dist) rep.min_dist = dist;
            if (dist > rep.max_dist) rep.max_dist = dist;
        }
        if (!g_quiet) {
            std::printf("-- Max dist : %f \n", rep.max_dist);
            std::printf("-- Min dist : %f \n", rep.min_dist);
        }
        for (int i = 0; i < nq; i++)
            if (idx2[2 * i] >= 0 && dist2[2 * i] <= std::max(2.5 * rep.min_dist, 0.35))
                rep.good_matches.push_back(DMatch{i, idx2[2 * i], dist2[2 * i]});
        if (!g_quiet) std::cout << "Good matches: " << rep.good_matches.size() << std::endl;
        rep.match_ms = ms_since(t0);
        finish_report(rep, img_object, pts_object, pts_scene, up_scale);
        return rep;
    }

    MatchReport Plot_Img_Matches_ocv(const mif::Image& img_object, const mif::Image&,
                                     const std::vector<Ipoint>& kp_object, const std::vector<Ipoint>& kp_scene,
                                     const bool up_scale)
    {
        MatchReport rep;
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<Point2f> pts_object, pts_scene;
        std::vector<float> desc_object, desc_scene;
        ConvertToKeyPoint(kp_object, pts_object, desc_object);
        ConvertToKeyPoint(kp_scene, pts_scene, desc_scene);
        std::vector<int> idx2;
        std::vector<float> dist2;
        const int nq = (int)kp_object.size(), nd = (int)kp_scene.size();
        if (!gpu_top2(desc_object, nq, desc_scene, nd, 64, idx2, dist2)) return rep;
        // common/kpmatch.h:289-297
        const float ratio_thresh = 0.75f;
        for (int i = 0; i < nq; i++)
            if (idx2[2 * i] >= 0 && dist2[2 * i] < ratio_thresh * dist2[2 * i + 1])
                rep.good_matches.push_back(DMatch{i, idx2[2 * i], dist2[2 * i]});
        if (!g_quiet) std::cout << "Found stable matches:\t" << rep.good_matches.size() << std::endl;
        rep.match_ms = ms_since(t0);
        if (!rep.good_matches.empty()) finish_report(rep, img_object, pts_object, pts_scene, up_scale);
        return rep;
    }

    bool write_report_json(const std::string& path, const MatchReport& rep)
    {
        return write_report_json(path, rep, Timing());
    }

    bool write_report_json(const std::string& path, const MatchReport& rep, const Timing& tm)
    {
        std::ofstream out(path);
        if (!out.is_open()) return false;
        out.precision(9);
        out << "{\n";
        if (tm.repeats > 0)
            out << "  \"timing_ms\": {\"repeats\": " << tm.repeats << ", \"extract_object\": " << tm.extract_object
                << ", \"extract_scene\": " << tm.extract_scene << ", \"convert\": " << tm.convert
                << ", \"match\": " << tm.match << ", \"homography\": " << tm.homography
                << ", \"total\": " << tm.total << ", \"total_min\": " << tm.total_min
                << ", \"feats_object\": " << tm.feats_object << ", \"feats_scene\": " << tm.feats_scene << "},\n";
        out << "  \"min_dist\": " << rep.min_dist << ", \"max_dist\": " << rep.max_dist << ",\n  \"good_matches\": [";
        for (size_t i = 0; i < rep.good_matches.size(); i++) {
            const DMatch& m = rep.good_matches[i];
            out << (i ? ", " : "") << "[" << m.queryIdx << ", " << m.trainIdx << ", " << m.distance << "]";
        }
        out << "],\n  \"has_homography\": " << (rep.has_homography ? "true" : "false") << ", \"inliers\": " << rep.inliers
            << ",\n  \"H\": [";
        for (int k = 0; k < 9; k++) out << (k ? ", " : "") << rep.H[k];
        out << "],\n  \"scene_corners\": [";
        for (int k = 0; k < 4; k++)
            out << (k ? ", " : "") << "[" << rep.scene_corners[k].x << ", " << rep.scene_corners[k].y << "]";
        out << "]\n}\n";
        return out.good();
    }
}
