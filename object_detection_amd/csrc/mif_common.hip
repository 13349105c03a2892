// Error reporting and device enumeration for libmi_features.so.
#include "mif_common.h"
#include "mif_math.h"

namespace {

// op: 0 expf, 1 exp2f, 2 sinf, 3 cosf, 4 atan2f(a, b), 5 expf_bounded
__global__ void k_math_probe(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                             int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c, r = 0.0f;
    switch (op) {
    case 0: r = mif_expf(a[i]); break;
    case 1: r = mif_exp2f(a[i]); break;
    case 2: mif_sincosf(a[i], &s, &c); r = s; break;
    case 3: mif_sincosf(a[i], &s, &c); r = c; break;
    case 4: r = mif_atan2f(a[i], b[i]); break;
    case 5: r = mif_expf_bounded(a[i]); break;
    }
    out[i] = r;
}

} // namespace

namespace mif {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

} // namespace mif

extern "C" {

const char* mif_last_error(void) { return mif::g_last_error.c_str(); }

int mif_math_probe(int device, int op, const float* a, const float* b, float* out, int n)
{
    if (op < 0 || op > 5 || n < 0 || !a || !out || (op == 4 && !b)) {
        mif::set_error("mif_math_probe: bad argument");
        return MIF_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return MIF_OK;
    mif::DeviceGuard guard(device);
    if (!guard.ok) { mif::set_error("mif_math_probe: cannot select device %d", device); return MIF_ERR_HIP; }
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    const size_t bytes = (size_t)n * sizeof(float);
    hipError_t e = hipMalloc(&da, bytes);
    if (e == hipSuccess) e = hipMalloc(&db, bytes);
    if (e == hipSuccess) e = hipMalloc(&dout, bytes);
    if (e == hipSuccess) e = hipMemcpy(da, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db, b ? b : a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_math_probe, dim3((n + 255) / 256), dim3(256), 0, nullptr, op, (const float*)da,
                           (const float*)db, dout, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    if (e != hipSuccess) { mif::set_error("mif_math_probe: %s", hipGetErrorString(e)); return MIF_ERR_HIP; }
    return MIF_OK;
}

int mif_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

} // extern "C"
