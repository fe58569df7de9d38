// Error reporting shared by the translation units that implement the C ABI (engine.hip, abi_ops.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace gitmi {
// formats the thread's message for gitmi_last_error() and returns 1 (the ABI's failure code)
int fail(const char* fmt, ...);
}  // namespace gitmi

#define HIPCK(expr)                                                                                   \
    do {                                                                                              \
        hipError_t e__ = (expr);                                                                      \
        if (e__ != hipSuccess) return gitmi::fail("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
    } while (0)
#define RCK(expr)                 \
    do {                          \
        int r__ = (expr);         \
        if (r__ != 0) return r__; \
    } while (0)
