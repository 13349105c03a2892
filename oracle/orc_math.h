// Math functions of the CPU checker (test infrastructure only).
//
// The default build takes exp / exp2 / atan / atan2 / sin / cos from the header the HIP kernels use
// (csrc/mif_math.h), which is what lets the parity tests ask for equal bits. The reference leaves these
// to whatever OpenCL device it runs on (built-in exp, atan2, sin, cos under -cl-fast-relaxed-math,
// SIFT/sift.cpp:1369-1370), so the checker's results are only as meaningful as they are independent of
// that choice. Built with -DORC_MATH_LIBM every one of them comes from the C library in double
// precision, rounded once to float, instead; tests/test_math_sensitivity.py runs both builds on the
// same images and states how far the features and matches move.
#pragma once
#include <cmath>
#include "../object_detection_amd/csrc/mif_math.h"

#ifdef ORC_MATH_LIBM
static inline float orc_libm_expf(float x) { return (float)std::exp((double)x); }
static inline float orc_libm_exp2f(float x) { return (float)std::exp2((double)x); }
static inline float orc_libm_atan2f(float y, float x) { return (float)std::atan2((double)y, (double)x); }
static inline float orc_libm_atan01(float a) { return (float)std::atan((double)a); }
static inline void orc_libm_sincosf(float x, float* s, float* c)
{
    *s = (float)std::sin((double)x);
    *c = (float)std::cos((double)x);
}
#define mif_expf orc_libm_expf
#define mif_expf_bounded orc_libm_expf
#define mif_exp2f orc_libm_exp2f
#define mif_atan2f orc_libm_atan2f
#define mif_atan01 orc_libm_atan01
#define mif_sincosf orc_libm_sincosf
#endif
