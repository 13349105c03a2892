// Does data written by one kernel come back faster when the next kernel reads it soon enough, i.e. does
// the 256 MB memory-side cache keep freshly written lines? Kernel A writes S bytes, kernel B reads them
// (sum into a sink); B is timed for several S. For comparison B is also timed after a 2 GiB buffer has
// been streamed through in between (nothing of S can be cached then).
//   hipcc --offload-arch=gfx950 -O3 -o mall_reuse mall_reuse.hip && ./mall_reuse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_write(float4* p, size_t n, float v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(v, v, v, v);
}
__global__ void k_read(const float4* p, size_t n, float* sink)
{
    float acc = 0.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) *sink = acc;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
int main()
{
    const size_t big = (size_t)2 << 30;
    float4 *buf, *flush;
    float* sink;
    CK(hipMalloc(&buf, big));
    CK(hipMalloc(&flush, big));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * 16, block = 256;
    printf("%10s %14s %14s\n", "S (MB)", "read soon GB/s", "read late GB/s");
    for (size_t mb : {16, 32, 64, 128, 192, 256, 512, 1024, 2048}) {
        const size_t n = mb * (1 << 20) / 16;
        float best[2] = {0, 0};
        for (int late = 0; late < 2; late++) {
            for (int rep = 0; rep < 5; rep++) {
                hipLaunchKernelGGL(k_write, dim3(grid), dim3(block), 0, 0, buf, n, (float)rep);
                if (late) hipLaunchKernelGGL(k_write, dim3(grid), dim3(block), 0, 0, flush, big / 16, 1.0f);
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k_read, dim3(grid), dim3(block), 0, 0, (const float4*)buf, n, sink);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const float gbs = (float)(mb * 1.048576e6 / (ms * 1e-3) / 1e9);
                if (gbs > best[late]) best[late] = gbs;
            }
        }
        printf("%10zu %14.0f %14.0f\n", mb, best[0], best[1]);
    }
    return 0;
}
