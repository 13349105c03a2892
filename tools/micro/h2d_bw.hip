// Host-to-device upload rate of one GPU: hipMemcpyAsync from pinned memory with 1..8 streams and several
// chunk sizes, pinned memory allocated with different flags, and a kernel that reads mapped host memory
// directly. Prints GB/s per variant.  hipcc --offload-arch=gfx950 -O2 -o h2d_bw h2d_bw.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_pull(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t bytes = (size_t)3 << 30;            // 3 GiB, about one batch of 512 BGR frames
    void* dev = nullptr;
    CK(hipMalloc(&dev, bytes));
    const unsigned flag_set[] = {hipHostMallocDefault, hipHostMallocNumaUser, hipHostMallocWriteCombined,
                                 hipHostMallocNonCoherent};
    const char* flag_name[] = {"default", "numa_user", "write_combined", "non_coherent"};
    for (int fi = 0; fi < 4; fi++) {
        void* host = nullptr;
        if (hipHostMalloc(&host, bytes, flag_set[fi]) != hipSuccess) { printf("%s: alloc failed\n", flag_name[fi]); (void)hipGetLastError(); continue; }
        memset(host, 1, bytes);
        for (int ns : {1, 2, 4, 8}) {
            for (size_t chunk : {(size_t)8 << 20, (size_t)64 << 20, bytes / ns}) {
                std::vector<hipStream_t> st(ns);
                for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
                double best = 0;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipDeviceSynchronize());
                    const double t0 = now();
                    size_t off = 0; int k = 0;
                    while (off < bytes) {
                        const size_t n = std::min(chunk, bytes - off);
                        CK(hipMemcpyAsync((char*)dev + off, (char*)host + off, n, hipMemcpyHostToDevice, st[k % ns]));
                        off += n; k++;
                    }
                    CK(hipDeviceSynchronize());
                    best = std::max(best, bytes / (now() - t0) / 1e9);
                }
                printf("%-15s streams %d chunk %6zu MiB : %6.1f GB/s\n", flag_name[fi], ns, chunk >> 20, best);
                for (auto& s : st) CK(hipStreamDestroy(s));
            }
        }
        void* mapped = nullptr;
        if (hipHostGetDevicePointer(&mapped, host, 0) == hipSuccess) {
            for (int blocks : {256, 1024, 4096}) {
                double best = 0;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipDeviceSynchronize());
                    const double t0 = now();
                    hipLaunchKernelGGL(k_pull, dim3(blocks), dim3(256), 0, nullptr, (const uint4*)mapped, (uint4*)dev, bytes / 16);
                    CK(hipDeviceSynchronize());
                    best = std::max(best, bytes / (now() - t0) / 1e9);
                }
                printf("%-15s kernel pull, %4d blocks        : %6.1f GB/s\n", flag_name[fi], blocks, best);
            }
        } else (void)hipGetLastError();
        CK(hipHostFree(host));
    }
    return 0;
}
