// Which compute units does a stream created with hipExtStreamCreateWithCUMask use? For a few masks, launches
// many short blocks on the masked stream and prints how many distinct (XCD, SE, SH, CU) places ran them, per XCD.
// hipcc --offload-arch=gfx950 -O2 -o cu_mask cu_mask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void k_where(unsigned* out, int spin)
{
    // HW_REG_HW_ID (4): cu_id [11:8], sh_id [12], se_id [15:13]; HW_REG_XCC_ID (20): xcc_id [3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
    float a = (float)threadIdx.x;
    for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;
    if (threadIdx.x == 0) out[blockIdx.x] = (hw & 0xff00u) | (xcc << 16) | (a == 12345.0f ? 1u : 0u);
}

static void run(const char* name, const std::vector<uint32_t>& mask)
{
    hipStream_t st;
    fprintf(stderr, "%s: creating the stream\n", name);
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        printf("%s: stream creation failed\n", name);
        return;
    }
    fprintf(stderr, "%s: stream created\n", name);
    const int blocks = 16384;
    unsigned* d = nullptr;
    if (hipMalloc(&d, blocks * sizeof(unsigned)) != hipSuccess) return;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_where, dim3(blocks), dim3(256), 0, st, d, 2000);
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(k_where, dim3(blocks), dim3(256), 0, st, d, 20000);
    (void)hipEventRecord(e1, st);
    (void)hipStreamSynchronize(st);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(blocks);
    (void)hipMemcpy(h.data(), d, blocks * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (unsigned v : h) per_xcc[v >> 16].insert((v >> 8) & 0xff);
    int total = 0;
    printf("%-28s %.2f ms  places per XCD:", name, ms);
    for (auto& kv : per_xcc) { printf(" x%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  total %d\n", total);
    (void)hipFree(d);
    (void)hipStreamDestroy(st);
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    fprintf(stderr, "compute units: %d\n", cus);
    std::vector<uint32_t> all(8, 0xffffffffu);
    run("all 256 bits", all);
    // (a mask of every other bit, 0x55555555 per word, made the runtime crash inside the stream creation on
    // ROCm 7.2: contiguous ranges only)
    std::vector<uint32_t> lo(8, 0), hi(8, 0), first64(8, 0);
    for (int i = 0; i < 4; i++) { lo[i] = 0xffffffffu; hi[4 + i] = 0xffffffffu; }
    first64[0] = first64[1] = 0xffffffffu;
    run("low 128 bits", lo);
    run("high 128 bits", hi);
    run("low 64 bits", first64);
    std::vector<uint32_t> lo96(8, 0);
    for (int i = 0; i < 3; i++) lo96[i] = 0xffffffffu;
    run("low 96 bits", lo96);
    return 0;
}
