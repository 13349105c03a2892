// TEST INFRASTRUCTURE ONLY -- CPU statement of the exact brute-force matcher
// that stands in for the reference's OpenCV FLANN calls
// (common/kpmatch.h:174-176 match, 284-286 knnMatch k=2).
//
// PARITY UNPINNED at this boundary: OpenCV 4.x is an un-vendored dependency of
// the reference (find_package(OpenCV), SIFT/CMakeLists.txt:5), FLANN is an
// approximate randomised search and no reference test pins its output. The
// definition checked here is exact L2 nearest neighbours; the tie rule (lowest
// database index wins) is the one of the reference's only brute-force matcher,
// SURF/coopcl_surf/CLSource/nearestNeighbor_kernel.cl:116-122 (strict '<').
//
//   nq_i = sum_k q_ik^2, nd_j = sum_k d_jk^2, dot_ij = sum_k q_ik d_jk  (fmaf chains, k ascending)
//   s_ij = fmaf(-2, dot_ij, nq_i + nd_j); the two smallest (s_ij, j) win
//   distance = sqrt(sum_k (q_ik - d_jk)^2) (fmaf chain); winners swapped if that orders them better
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>
#include <algorithm>

namespace {

struct Top2 {
    float s1, s2;
    int i1, i2;
};

inline bool before(float sa, int ia, float sb, int ib) { return sa < sb || (sa == sb && ia < ib); }

inline void insert(Top2& t, float s, int i)
{
    if (before(s, i, t.s1, t.i1)) {
        t.s2 = t.s1; t.i2 = t.i1; t.s1 = s; t.i1 = i;
    } else if (before(s, i, t.s2, t.i2)) {
        t.s2 = s; t.i2 = i;
    }
}

float exact_dist(const float* q, const float* d, int dim)
{
    float acc = 0.0f;
    for (int k = 0; k < dim; k++) {
        const float df = q[k] - d[k];
        acc = fmaf(df, df, acc);
    }
    return sqrtf(acc);
}

} // namespace

extern "C" {

int orc_match_top2_scores(const float* q, int nq, const float* db, int nd, int dim, int ld, int32_t* idx2,
                          float* dist2, float* score2);

int orc_match_top2(const float* q, int nq, const float* db, int nd, int dim, int ld, int32_t* idx2, float* dist2)
{
    return orc_match_top2_scores(q, nq, db, nd, dim, ld, idx2, dist2, nullptr);
}

// Same contract as mif_match_top2_scores of include/mi_features.h.
int orc_match_top2_scores(const float* q, int nq, const float* db, int nd, int dim, int ld, int32_t* idx2,
                          float* dist2, float* score2)
{
    const float inf = std::numeric_limits<float>::infinity();
    std::vector<float> dn((size_t)std::max(nd, 1)), dT((size_t)dim * std::max(nd, 1));
    for (int j = 0; j < nd; j++) {
        float acc = 0.0f;
        for (int k = 0; k < dim; k++) {
            const float v = db[(size_t)j * ld + k];
            acc = fmaf(v, v, acc);
            dT[(size_t)k * nd + j] = v;
        }
        dn[j] = acc;
    }
    constexpr int CH = 2048;
#pragma omp parallel
    {
        std::vector<float> acc(CH);
#pragma omp for schedule(dynamic, 4)
        for (int i = 0; i < nq; i++) {
            const float* qr = q + (size_t)i * ld;
            float qn = 0.0f;
            for (int k = 0; k < dim; k++) qn = fmaf(qr[k], qr[k], qn);
            Top2 t{inf, inf, 0x7fffffff, 0x7fffffff};
            for (int j0 = 0; j0 < nd; j0 += CH) {
                const int n = std::min(CH, nd - j0);
                for (int j = 0; j < n; j++) acc[j] = 0.0f;
                for (int k = 0; k < dim; k++) {
                    const float qk = qr[k];
                    const float* col = &dT[(size_t)k * nd + j0];
                    for (int j = 0; j < n; j++) acc[j] = fmaf(qk, col[j], acc[j]);
                }
                for (int j = 0; j < n; j++) insert(t, fmaf(-2.0f, acc[j], qn + dn[j0 + j]), j0 + j);
            }
            int j1 = t.i1 < nd ? t.i1 : -1, j2 = t.i2 < nd ? t.i2 : -1;
            float e1 = j1 >= 0 ? exact_dist(qr, db + (size_t)j1 * ld, dim) : inf;
            float e2 = j2 >= 0 ? exact_dist(qr, db + (size_t)j2 * ld, dim) : inf;
            float s1 = j1 >= 0 ? t.s1 : inf, s2 = j2 >= 0 ? t.s2 : inf;
            if (j2 >= 0 && (e2 < e1 || (e2 == e1 && j2 < j1))) {
                std::swap(e1, e2);
                std::swap(j1, j2);
                std::swap(s1, s2);
            }
            idx2[2 * i] = j1; idx2[2 * i + 1] = j2;
            dist2[2 * i] = e1; dist2[2 * i + 1] = e2;
            if (score2) { score2[2 * i] = s1; score2[2 * i + 1] = s2; }
        }
    }
    return 0;
}

} // extern "C"
