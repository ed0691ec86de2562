// Shared host-side helpers for the PULSE gfx950 library (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "pulse_hip.h"

namespace pulse {

char* last_error_buf();  // thread-local, 512 bytes (defined in capi.cpp)

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PULSE_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return PULSE_OK;
}

inline hipStream_t as_stream(pulse_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;  // gfx950 wavefront

}  // namespace pulse

#define PULSE_REQUIRE(cond, ...) \
    do { if (!(cond)) return ::pulse::fail(PULSE_ERR_INVALID_ARG, __VA_ARGS__); } while (0)
