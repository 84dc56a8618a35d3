// mpcqp_hostutil.h -- small host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/mpcqp.h"

extern thread_local std::string g_hip_err;     // text behind MPCQP_ERR_DEVICE (mpcqp_last_hip_error)

#define HIPCHK(expr)                                                                 \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            g_hip_err = std::string(#expr) + ": " + hipGetErrorString(e_);           \
            return MPCQP_ERR_DEVICE;                                                 \
        }                                                                            \
    } while (0)

// Every entry point selects the handle's device; the caller's current device is restored when the
// entry point returns (a torch process, or a Julia host driving several handles, keeps its own).
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
#define ON_DEVICE(h)                                                                   \
    DeviceGuard guard_((h)->device);                                                   \
    if (!guard_.ok) { g_hip_err = "hipSetDevice failed"; return MPCQP_ERR_DEVICE; }

