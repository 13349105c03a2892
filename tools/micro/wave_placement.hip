// Which SIMD does each wave of a workgroup land on? Prints, for a few workgroups of 256 and 512 threads,
// the SIMD and compute-unit ids of every wave (HW_ID register).  hipcc --offload-arch=gfx950 -O2 -o wave_placement wave_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_where(unsigned* out)
{
    // HW_REG_HW_ID (4): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    const unsigned hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = hw;
}

int main()
{
    for (int threads : {256, 512}) {
        const int blocks = 6, waves = threads / 64;
        unsigned* d = nullptr;
        if (hipMalloc(&d, blocks * waves * sizeof(unsigned)) != hipSuccess) return 1;
        hipLaunchKernelGGL(k_where, dim3(blocks), dim3(threads), 0, nullptr, d);
        std::vector<unsigned> h(blocks * waves);
        if (hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        for (int b = 0; b < blocks; b++) {
            printf("%d threads, block %d: ", threads, b);
            for (int w = 0; w < waves; w++) {
                const unsigned v = h[b * waves + w];
                printf("w%d simd%u cu%u se%u slot%u | ", w, (v >> 4) & 3, (v >> 8) & 15, (v >> 13) & 7, v & 15);
            }
            printf("\n");
        }
        (void)hipFree(d);
    }
    return 0;
}
