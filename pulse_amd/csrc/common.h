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

// ---- the exact three-way bf16 split of fp32 values (gemm_x3 / gemm_x3p arithmetic): x = p0 + p1 + p2, p0 = bf16(x), p1 = bf16(x - p0),
// p2 = bf16(x - p0 - p1), round to nearest even, the remainders exact.  Pairs of elements, packed {hi, lo} per dword.
typedef float pulse_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pulse_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned split_pack_rn(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((pulse_f32x2){lo, hi}, pulse_bf16x2));
}
__device__ __forceinline__ float split_bitsf(unsigned v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ void split_pair3(float a, float b, unsigned& q0, unsigned& q1, unsigned& q2) {
    q0 = split_pack_rn(a, b);
    const float ra = a - split_bitsf(q0 << 16), rb = b - split_bitsf(q0 & 0xffff0000u);
    q1 = split_pack_rn(ra, rb);
    const float sa = ra - split_bitsf(q1 << 16), sb = rb - split_bitsf(q1 & 0xffff0000u);
    q2 = split_pack_rn(sa, sb);
}

}  // namespace pulse

#define PULSE_REQUIRE(cond, ...) \
    do { if (!(cond)) return ::pulse::fail(PULSE_ERR_INVALID_ARG, __VA_ARGS__); } while (0)
