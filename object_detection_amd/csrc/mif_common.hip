// Error reporting and device enumeration for libmi_features.so.
#include "mif_common.h"

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

int mif_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

} // extern "C"
