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

// ---- workgroup -> (output tile, batch slot, k split), XCD-aware ---------------------------------------------------------------------------
// Workgroups are dealt to the 8 XCDs round-robin in linear launch order (x fastest), and each XCD has its own L2.
//  * no split-K: every XCD owns a contiguous band of the output tiles (an A row panel is fetched by one XCD only; B by all eight);
//  * split-K (the weight-gradient launches: small outputs, long reductions): every XCD owns a K RANGE instead -- split s runs on XCD s (or on
//    8 / S XCDs, each with a contiguous share of the tiles).  The workgroups of an XCD then walk the same k rows of A and B in step, so every
//    operand element crosses the fabric once per XCD that needs it instead of once per XCD: the layer-1 weight gradient of cfg2 (4 splits) goes
//    from |dZ| + 8 |X| to |dZ| + 2 |X| of fetch (624 -> 256 MB per launch).  Falls back to the band order when the counts do not divide.
#ifndef PULSE_SPLITK_XCD
#define PULSE_SPLITK_XCD 1
#endif
struct WgMap { int id, bz, sp; };
__device__ __forceinline__ WgMap map_workgroup(int ntile, int batch, int splitk) {
    const int cnt = ntile * batch;                                   // (tile, batch slot) pairs per split
    if (PULSE_SPLITK_XCD && splitk > 1) {
        const int L = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = L & 7, q = L >> 3;
        if ((splitk & 7) == 0 && ((cnt * splitk) & 7) == 0) {        // 8, 16, 32 splits: split = xcd + 8 * (q % (S / 8))
            const int g8 = splitk >> 3;
            const int sp = xcd + 8 * (q % g8), rest = q / g8;
            return WgMap{rest % ntile, rest / ntile, sp};
        }
        if ((8 % splitk) == 0 && (cnt % (8 / splitk)) == 0) {        // 2, 4 splits: 8 / S XCDs per split, contiguous shares of the tiles
            const int per = 8 / splitk, share = cnt / per;
            const int sp = xcd / per, rest = (xcd % per) * share + q;
            return WgMap{rest % ntile, rest / ntile, sp};
        }
    }
    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int q8 = ntile >> 3, rr = ntile & 7;
    const int id = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + loc;
    const int z = blockIdx.y;
    return WgMap{id, z / splitk, z % splitk};
}

long long* gemm_debug_buffer();  // the calling thread's pulse_gemm_set_debug_buffer pointer (gemm_f32.hip)
int gemm_option(int key);        // the calling thread's pulse_gemm_set_option value (gemm_f32.hip)

}  // namespace pulse

#define PULSE_REQUIRE(cond, ...) \
    do { if (!(cond)) return ::pulse::fail(PULSE_ERR_INVALID_ARG, __VA_ARGS__); } while (0)
