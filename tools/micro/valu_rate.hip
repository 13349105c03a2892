// Issue rate of scalar-lane and packed fp32 FMA (and a few other vector ops) with 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float a[8];
    v2f p[8];
    double d[4];
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; p[i] = (v2f){a[i], a[i] + 1.0f}; }
    for (int i = 0; i < 4; i++) d[i] = a[i];
    const float m = 0.999f, c = 0.001f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);
                if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], (v2f){m, m}, (v2f){c, c});
                if (MODE == 2) a[i] = (float)(int)a[i] * m + c;                 // cvt, cvt, fma-ish
                if (MODE == 3) d[i & 3] = __builtin_fma(d[i & 3], (double)m, (double)c);
                if (MODE == 4) a[i] = (float)((double)a[i] * 0.999);            // cvt f64, mul f64, cvt f32
            }
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    for (int i = 0; i < 4; i++) s += (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, int cus, double ops_per_inner)
{
    float* out;
    const int blocks = blocks_per_cu * cus;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD per second -> cycles per wave-instruction at 2.4 GHz
    const double winst = (double)blocks * 4 * iters * 64 * ops_per_inner;
    const double per_simd_per_s = winst / (ms * 1e-3) / (cus * 4.0);
    printf("%-28s waves/SIMD %d: %.2f cycles per wave-instruction (at 2.4 GHz)\n", name, blocks_per_cu, 2.4e9 / per_simd_per_s);
    (void)hipFree(out);
}

int main()
{
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w, cus, 1);
        run<1>("v_pk_fma_f32", w, cus, 1);
        run<3>("v_fma_f64", w, cus, 1);
    }
    run<2>("cvt_i32 + cvt_f32 + fma (3)", 4, cus, 3);
    run<4>("cvt_f64 + mul_f64 + cvt_f32 (3)", 4, cus, 3);
    return 0;
}
