// tvz_common.h — shared host-side helpers for libtvz.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "tvz.h"

#define TVZ_EXPORT extern "C" __attribute__((visibility("default")))

namespace tvz {

char *err_buf();  // thread-local, 512 bytes

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define TVZ_HIP(expr)                                                                     \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return tvz::fail(TVZ_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                 \
                             hipGetErrorString(_e), __FILE__, __LINE__);                  \
    } while (0)

#define TVZ_REQUIRE(cond, ...)                                                            \
    do {                                                                                  \
        if (!(cond)) return tvz::fail(TVZ_ERR_INVALID, __VA_ARGS__);                      \
    } while (0)

inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace tvz
