// Shared host-side helpers for the cookietts HIP library (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#include "../../include/cookietts_hip.h"

namespace ctts {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

#define CTTS_CHECK_ARG(cond, ...)            \
    do {                                     \
        if (!(cond)) {                       \
            ::ctts::set_error(__VA_ARGS__);  \
            return CTTS_E_ARG;               \
        }                                    \
    } while (0)

#define CTTS_CHECK_LAUNCH(what)                                                          \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            ::ctts::set_error("%s: %s", what, hipGetErrorString(e__));                   \
            return CTTS_E_LAUNCH;                                                        \
        }                                                                                \
    } while (0)

#define CTTS_CHECK_HIP(expr)                                                             \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) {                                                         \
            ::ctts::set_error("%s: %s", #expr, hipGetErrorString(e__));                  \
            return CTTS_E_LAUNCH;                                                        \
        }                                                                                \
    } while (0)

}  // namespace ctts
