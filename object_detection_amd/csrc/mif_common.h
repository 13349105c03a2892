// Internal helpers shared by the translation units of libmi_features.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <string>
#include "../../include/mi_features.h"

namespace mif {

void set_error(const char* fmt, ...);

// Evaluates a HIP call; on failure records the message and returns MIF_ERR_HIP
// from the enclosing function.
#define MIF_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            ::mif::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,  \
                             __LINE__);                                                        \
            return MIF_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

// Makes `device` current for the lifetime of the object and restores the
// previous device afterwards.
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (hipSetDevice(device) == hipSuccess);
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

} // namespace mif
