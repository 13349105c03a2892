// Does the LDS unit's in-place float add (ds_add_f32) give the bits of the vector unit's add, denormal
// operands and results included? Every lane adds the same pseudo-random sequence (normal, tiny and denormal
// magnitudes, both signs) to one LDS word with the atomic and to a register with `+`; the host compares.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o lds_add lds_add.hip && ./lds_add
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k(const float* __restrict__ vals, int n, float* __restrict__ out_lds, float* __restrict__ out_reg)
{
    __shared__ float s[256];
    const int t = threadIdx.x;
    s[t] = 0.0f;
    float r = 0.0f;
    __syncthreads();
    for (int i = 0; i < n; i++) {
        const float v = vals[(size_t)i * 256 + t];
        __hip_atomic_fetch_add(&s[t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        r += v;
    }
    __syncthreads();
    out_lds[blockIdx.x * 256 + t] = s[t];
    out_reg[blockIdx.x * 256 + t] = r;
}

int main()
{
    const int n = 4096;
    std::vector<float> h((size_t)n * 256);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (size_t i = 0; i < h.size(); i++) {
        const uint64_t r = rnd();
        uint32_t bits;
        switch (r & 3) {
        case 0: bits = (uint32_t)(r >> 32) & 0x807fffffu; break;                       // denormal
        case 1: bits = ((uint32_t)(r >> 32) & 0x807fffffu) | (uint32_t)((1u + (r >> 8) % 4) << 23); break;   // just above
        default: bits = ((uint32_t)(r >> 32) & 0x807fffffu) | ((100u + (r >> 8) % 40) << 23); break;
        }
        // a lane class per column: columns 0..63 see only denormal / tiny values (sums stay denormal)
        if ((i % 256) < 64) bits &= 0x807fffffu | (1u << 23);
        std::memcpy(&h[i], &bits, 4);
    }
    float *d, *a, *b;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&a, 1024);
    hipMalloc(&b, 1024);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, n, a, b);
    std::vector<float> ha(256), hb(256);
    if (hipMemcpy(ha.data(), a, 1024, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    hipMemcpy(hb.data(), b, 1024, hipMemcpyDeviceToHost);
    int bad = 0, denorm_results = 0;
    for (int t = 0; t < 256; t++) {
        uint32_t x, y;
        std::memcpy(&x, &ha[t], 4);
        std::memcpy(&y, &hb[t], 4);
        bad += x != y;
        denorm_results += (y & 0x7f800000u) == 0 && (y & 0x7fffffu) != 0;
    }
    printf("lds add vs vector add: %d of 256 sums differ (%d sums are denormal)\n", bad, denorm_results);
    return bad != 0;
}
