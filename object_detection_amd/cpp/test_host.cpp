// CPU-only checks of the host layer (no GPU call): argument parsing, image I/O, the shifted
// descriptor rows and the RANSAC homography. Exit code 0 = all passed.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <random>

#include "ccl_surf.h"
#include "kpmatch.h"
#include "parse_args.h"
#include "sift.h"

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static bool parse(std::vector<const char*> v, iargs::app_args& a)
{
    return a.parse((int)v.size(), const_cast<char**>(v.data()));
}

int main(int argc, char** argv)
{
    {   // common/parse_args.h:32-109
        iargs::app_args a;
        CHECK(parse({"app", "-m", "a.pgm", "b.pgm", "-f", "0.5"}, a) && a.app_mode == "-m" && a.img_path_b == "b.pgm" && a.offload == 0.5f);
        iargs::app_args b;
        CHECK(!parse({"app", "-m", "a.pgm", "b.pgm"}, b));                     // -f is mandatory
        iargs::app_args c;
        CHECK(!parse({"app", "-i", "a.pgm", "-f", "1.5"}, c));                 // offload out of range
        iargs::app_args d;
        CHECK(parse({"app", "-v", "x", "-f", "-1"}, d) && d.offload == -1.0f); // -1 = auto
        iargs::app_args e;
        CHECK(parse({"app", "-h"}, e));
        iargs::app_args f;
        CHECK(!parse({"app", "-f", "0"}, f));                                  // no mode
    }
    {   // image round trip, gray and colour
        mif::Image im;
        im.rows = 5; im.cols = 7; im.nchannels = 3;
        im.storage.resize(5 * 7 * 3);
        for (size_t i = 0; i < im.storage.size(); i++) im.storage[i] = (unsigned char)(i * 7 + 3);
        im.data = im.storage.data();
        const std::string dir = argc > 1 ? argv[1] : "/tmp";
        CHECK(mif::imwrite(dir + "/mif_t.ppm", im));
        const auto back = mif::imread(dir + "/mif_t.ppm", false);
        CHECK(back.rows == 5 && back.cols == 7 && back.channels() == 3 && back.storage == im.storage);
        const auto gray = mif::imread(dir + "/mif_t.ppm", true);
        CHECK(gray.channels() == 1 && gray.storage.size() == 35);
        CHECK(mif::imread(dir + "/does_not_exist.pgm", true).empty());
        mif::FrameReader none(dir + "/nope_%d.pgm");
        CHECK(!none.isOpened());
    }
    {   // common/kpmatch.h:15,127-128,141-142: matrix row = [0, d0 .. d126]
        std::vector<SIFT::sfeature> feats(2);
        feats[0] = SIFT::sfeature{1.f, 2.f, 1.6f, 2, 1};
        feats[1] = SIFT::sfeature{3.f, 4.f, 2.0f, 1, 0};
        std::vector<float> ori = {0.5f, -1.0f}, desc(256);
        for (int i = 0; i < 256; i++) desc[i] = (float)(i + 1);
        std::vector<fmatches::sift_feat> kp;
        fmatches::ConvertToSift_feat<SIFT::sfeature>(feats, ori, desc, kp);
        CHECK(kp.size() == 2 && kp[0].descriptor.size() == 129 && kp[0].descriptor[0] == 0.0f && kp[0].descriptor[1] == 1.0f);
        CHECK(kp[1].octave_id == 0.0f && kp[1].scale == 1.0f && kp[1].orient == -1.0f);
        std::vector<fmatches::Point2f> pts;
        std::vector<float> mat;
        fmatches::ConvertToKeyPoint(kp, pts, mat);
        CHECK(mat.size() == 256 && mat[0] == 0.0f && mat[1] == 1.0f && mat[127] == 127.0f && mat[128] == 0.0f && mat[129] == 129.0f);
    }
    {   // homography from noisy correspondences with 30 % outliers
        const double Ht[9] = {0.9, -0.2, 30, 0.15, 1.1, -12, 1e-4, -2e-4, 1};
        std::mt19937 rng(7);
        std::uniform_real_distribution<float> u(0.f, 400.f), noise(-0.3f, 0.3f);
        std::vector<fmatches::Point2f> a, b;
        for (int i = 0; i < 120; i++) {
            fmatches::Point2f p{u(rng), u(rng)};
            fmatches::Point2f q = fmatches::perspectiveTransform(p, Ht);
            if (i % 10 < 3) q = fmatches::Point2f{u(rng), u(rng)};
            else { q.x += noise(rng); q.y += noise(rng); }
            a.push_back(p); b.push_back(q);
        }
        double H[9];
        std::vector<char> mask;
        CHECK(fmatches::findHomographyRansac(a, b, H, &mask));
        int inl = 0;
        for (char m : mask) inl += m;
        CHECK(inl >= 80 && inl <= 95);
        for (const auto& p : {fmatches::Point2f{0, 0}, fmatches::Point2f{400, 0}, fmatches::Point2f{400, 400}, fmatches::Point2f{0, 400}}) {
            const auto q0 = fmatches::perspectiveTransform(p, Ht), q1 = fmatches::perspectiveTransform(p, H);
            CHECK(std::hypot(q0.x - q1.x, q0.y - q1.y) < 1.0);
        }
        double H2[9];
        CHECK(fmatches::findHomographyRansac(a, b, H2));                       // seeded: same model again
        for (int k = 0; k < 9; k++) CHECK(H[k] == H2[k]);
        std::vector<fmatches::Point2f> few(a.begin(), a.begin() + 3), few2(b.begin(), b.begin() + 3);
        CHECK(!fmatches::findHomographyRansac(few, few2, H));
    }
    {   // interface behaviour that needs no GPU
        size_t ms = 0;
        std::vector<SIFT::sfeature> feats;
        unsigned char px[16] = {0};
        CHECK(SIFT::extract_features(ms, 1.5f, px, 4, 4, 1, feats) != 0);      // offload out of range
        CHECK(ccl_surf::run(mif::Image::wrap(px, 4, 4, 1)).empty());           // driver not initialised
        CHECK(!ccl_surf::initialize_surf_driver({{"x.cl", "not_a_kernel"}}));
        CHECK(!ccl_surf::initialize_surf_driver({}, false, 2.0f));
        CHECK(sizeof(ccl_surf::Ipoint) == 280 && sizeof(fmatches::Ipoint) == 280);
    }
    std::cout << "host tests passed" << std::endl;
    return 0;
}
