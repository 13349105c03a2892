// Sustained rate of v_mfma_f32_32x32x2_f32 with nothing else in the loop:
// waves per SIMD x independent accumulation chains. Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0)
{
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++)
        for (int r = 0; r < 16; r++) acc[c][r] = 0.0f;
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.0f;
    for (int c = 0; c < CHAINS; c++)
        for (int r = 0; r < 16; r++) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
void run(int blocks_per_cu, int cus)
{
    float* out;
    const int blocks = blocks_per_cu * cus;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, 2.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 16 * CHAINS * 4096.0;
    printf("chains %d  blocks/CU %d (waves/SIMD %d): %.1f TFLOP/s\n", CHAINS, blocks_per_cu, blocks_per_cu, flops / ms / 1e9);
    hipFree(out);
}

int main()
{
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs %d\n", cus);
    run<1>(1, cus); run<2>(1, cus); run<4>(1, cus);
    run<1>(2, cus); run<2>(2, cus); run<4>(2, cus);
    run<2>(4, cus);
    return 0;
}
