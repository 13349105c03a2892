// Elementary functions shared by the HIP kernels and by CPU checkers.
//
// Every function here is built only from IEEE-754 single/double add, sub, mul,
// div, fma, floor and integer bit operations, so a gcc/x86-64 build and a
// hipcc/gfx950 build return bit-identical results as long as both are compiled
// with -ffp-contract=off (all fused operations are spelled out with fmaf).
// The reference computes these with OpenCL native_* built-ins under
// -cl-fast-relaxed-math (SIFT/sift.cpp:1369-1370), whose bits nothing pins; a
// single fixed implementation is what makes keypoint counts, histogram bins and
// descriptors reproducible across CPU and GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MIF_HD __host__ __device__ __forceinline__
#else
#define MIF_HD static inline
#endif

#define MIF_PI_D 3.14159265358979323846
#define MIF_PI_F 3.14159274101257324219f
#define MIF_PIO2_F 1.57079637050628662109f
#define MIF_PIO4_F 0.78539818525314331055f

MIF_HD float mif_bits_to_float(uint32_t u) { return __builtin_bit_cast(float, u); }
MIF_HD uint32_t mif_float_to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

MIF_HD float mif_fabsf(float x) { return mif_bits_to_float(mif_float_to_bits(x) & 0x7fffffffu); }

// round half away from zero (OpenCL round()); exact for |x| < 2^23.
MIF_HD float mif_roundf(float x)
{
    const float a = mif_fabsf(x);
    const float r = __builtin_floorf(a + 0.5f);
    return x < 0.0f ? -r : r;
}

MIF_HD double mif_round(double x)
{
    const double a = x < 0.0 ? -x : x;
    const double r = __builtin_floor(a + 0.5);
    return x < 0.0 ? -r : r;
}

// 2^n for integer n in [-126, 127].
MIF_HD float mif_pow2i(int n) { return mif_bits_to_float((uint32_t)(n + 127) << 23); }

// e^r for |r| <= 0.35 : degree-7 Taylor polynomial, Horner form with fma.
MIF_HD float mif_exp_poly(float r)
{
    float p = 1.98412698412698412698e-4f;            // 1/5040
    p = __builtin_fmaf(p, r, 1.38888888888888888889e-3f); // 1/720
    p = __builtin_fmaf(p, r, 8.33333333333333333333e-3f); // 1/120
    p = __builtin_fmaf(p, r, 4.16666666666666666667e-2f); // 1/24
    p = __builtin_fmaf(p, r, 1.66666666666666666667e-1f); // 1/6
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    return p;
}

// e^x, about 1 ulp. Underflows to 0 below -87, saturates to +inf above 88.
MIF_HD float mif_expf(float x)
{
    if (!(x >= -87.0f)) return (x != x) ? x : 0.0f;
    if (x > 88.0f) return mif_bits_to_float(0x7f800000u);
    const float n = __builtin_floorf(__builtin_fmaf(x, 1.44269504088896340736f, 0.5f));
    float r = __builtin_fmaf(n, -0.693145751953125f, x);          // ln2 high part (12 bits)
    r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);        // ln2 low part
    return mif_exp_poly(r) * mif_pow2i((int)n);
}

// mif_expf for callers that know -87 <= x <= 88: the same operations without the range
// guards, hence the same result on that interval.
MIF_HD float mif_expf_bounded(float x)
{
    const float n = __builtin_floorf(__builtin_fmaf(x, 1.44269504088896340736f, 0.5f));
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
    // scaling by 2^n is exact either way (n >= -126, the polynomial is near 1); ldexp is one GPU instruction
    return __builtin_ldexpf(mif_exp_poly(r), (int)n);
}

// 2^x for |x| < 126, about 1 ulp.
MIF_HD float mif_exp2f(float x)
{
    const float n = __builtin_floorf(x + 0.5f);
    const float f = x - n;                                        // exact, |f| <= 0.5
    return mif_exp_poly(f * 0.693147180559945309417f) * mif_pow2i((int)n);
}

// atan(a) for a in [0, 1].
MIF_HD float mif_atan01(float a)
{
    float y0 = 0.0f;
    if (a > 0.4142135679721832275390625f) {                       // tan(pi/8)
        y0 = MIF_PIO4_F;
        a = (a - 1.0f) / (a + 1.0f);
    }
    const float z = a * a;
    float p = 8.05374449538e-2f;
    p = __builtin_fmaf(p, z, -1.38776856032e-1f);
    p = __builtin_fmaf(p, z, 1.99777106478e-1f);
    p = __builtin_fmaf(p, z, -3.33329491539e-1f);
    p = __builtin_fmaf(p * z, a, a);
    return y0 + p;
}

// atan2(y, x) in [-pi, pi]; atan2(0, 0) = 0. One division: with mn <= mx the magnitudes,
// atan(mn / mx) = atan(a) for a = mn / mx <= tan(pi/8), and pi/4 + atan((mn - mx) / (mn + mx))
// above, so the reduced argument is formed before dividing.
MIF_HD float mif_atan2f(float y, float x)
{
    const float ax = mif_fabsf(x), ay = mif_fabsf(y);
    const float mx = ax > ay ? ax : ay;
    const float mn = ax > ay ? ay : ax;
    if (!(mx > 0.0f)) return 0.0f;
    const bool upper = mn > 0.4142135679721832275390625f * mx;    // mn / mx > tan(pi/8)
    const float num = upper ? mn - mx : mn;
    const float den = upper ? mn + mx : mx;
    const float a = num / den;                                    // |a| <= tan(pi/8)
    const float z = a * a;
    float p = 8.05374449538e-2f;
    p = __builtin_fmaf(p, z, -1.38776856032e-1f);
    p = __builtin_fmaf(p, z, 1.99777106478e-1f);
    p = __builtin_fmaf(p, z, -3.33329491539e-1f);
    p = __builtin_fmaf(p * z, a, a);
    float r = upper ? MIF_PIO4_F + p : p;
    if (ay > ax) r = MIF_PIO2_F - r;
    if (x < 0.0f) r = MIF_PI_F - r;
    return y < 0.0f ? -r : r;
}

// sin and cos of x for |x| <= 64, about 1 ulp.
MIF_HD void mif_sincosf(float x, float* s_out, float* c_out)
{
    const float ax = mif_fabsf(x);
    int j = (int)(ax * 1.27323954473516268615f);                  // 4/pi
    j += (j & 1);
    const float y = (float)j;
    float r = __builtin_fmaf(y, -0.78515625f, ax);                // pi/4 in three parts
    r = __builtin_fmaf(y, -2.4187564849853515625e-4f, r);
    r = __builtin_fmaf(y, -3.77489497744594108e-8f, r);
    const float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = __builtin_fmaf(ps, z, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    ps = __builtin_fmaf(ps * z, r, r);
    float pc = 2.443315711809948e-5f;
    pc = __builtin_fmaf(pc, z, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    pc = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    const int q = (j >> 1) & 3;                                   // quadrant of |x|
    float s = (q & 1) ? pc : ps;
    float c = (q & 1) ? ps : pc;
    if (q == 1 || q == 2) c = -c;
    if (q >= 2) s = -s;
    if (x < 0.0f) s = -s;
    *s_out = s;
    *c_out = c;
}
