// Stand-alone batched rotation ops (one row per thread).
//
// These are the per-op entry points of the C ABI (include/pulse_hip.h section 1);
// the fused env-step kernel (env_step.hip) inlines the same device functions.
// Memory-bound elementwise work: rows are 12-28 bytes, so a wave reads a contiguous
// 768 B - 1.8 KB span per instruction; the AoS stride is left to the L1/TA to merge
// (these entry points exist for API completeness and parity tests -- the hot path
// uses the fused kernel, which stages through LDS).
// Compiled with -ffp-contract=off.
#include "common.h"
#include "rot_math.h"

namespace pulse {

enum RotOp { OP_QMUL, OP_QCONJ, OP_QROT, OP_Q2AA, OP_Q2EXP, OP_Q2TN, OP_EXP2Q, OP_SLERP, OP_HEADING, OP_HEADQ, OP_HEADQ_INV };

__device__ __forceinline__ Q4 ldq(const float* p, int64_t i) { return Q4{p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]}; }
__device__ __forceinline__ V3 ldv(const float* p, int64_t i) { return V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ void stq(float* p, int64_t i, Q4 q) { p[4 * i] = q.x; p[4 * i + 1] = q.y; p[4 * i + 2] = q.z; p[4 * i + 3] = q.w; }
__device__ __forceinline__ void stv(float* p, int64_t i, V3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }

template <int OP>
__global__ void __launch_bounds__(256) rot_op_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    const float* __restrict__ c, float* __restrict__ o0,
                                                    float* __restrict__ o1, int64_t m) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        if constexpr (OP == OP_QMUL) {
            stq(o0, i, qmul(ldq(a, i), ldq(b, i)));
        } else if constexpr (OP == OP_QCONJ) {
            stq(o0, i, qconj(ldq(a, i)));
        } else if constexpr (OP == OP_QROT) {
            stv(o0, i, qrot(ldq(a, i), ldv(b, i)));
        } else if constexpr (OP == OP_Q2AA) {
            V3 ax;
            const float ang = q_to_angle_axis(ldq(a, i), &ax);
            o0[i] = ang;
            stv(o1, i, ax);
        } else if constexpr (OP == OP_Q2EXP) {
            V3 ax;
            const float ang = q_to_angle_axis(ldq(a, i), &ax);
            stv(o0, i, V3{ang * ax.x, ang * ax.y, ang * ax.z});
        } else if constexpr (OP == OP_Q2TN) {
            float t[6];
            q_to_tan_norm(ldq(a, i), t);
#pragma unroll
            for (int k = 0; k < 6; ++k) o0[6 * i + k] = t[k];
        } else if constexpr (OP == OP_EXP2Q) {
            stq(o0, i, exp_map_to_q(ldv(a, i)));
        } else if constexpr (OP == OP_SLERP) {
            stq(o0, i, slerp(ldq(a, i), ldq(b, i), c[i]));
        } else if constexpr (OP == OP_HEADING) {
            o0[i] = heading_angle(ldq(a, i));
        } else if constexpr (OP == OP_HEADQ) {
            stq(o0, i, heading_quat(ldq(a, i), false));
        } else if constexpr (OP == OP_HEADQ_INV) {
            stq(o0, i, heading_quat(ldq(a, i), true));
        }
    }
}

template <int OP>
static int launch(const char* name, const float* a, const float* b, const float* c, float* o0, float* o1, int64_t m,
                  pulse_stream_t s) {
    PULSE_REQUIRE(m >= 0, "%s: negative row count", name);
    if (m == 0) return PULSE_OK;
    PULSE_REQUIRE(a != nullptr && o0 != nullptr, "%s: null pointer", name);
    const int block = 256;
    int64_t grid = (m + block - 1) / block;
    if (grid > 2048) grid = 2048;  // 256 CUs x 8 blocks; grid-stride the rest
    hipLaunchKernelGGL(rot_op_kernel<OP>, dim3((unsigned)grid), dim3(block), 0, as_stream(s), a, b, c, o0, o1, m);
    return check_launch(name);
}

}  // namespace pulse

using namespace pulse;

extern "C" {
int pulse_quat_mul(const float* a, const float* b, float* out, int64_t m, pulse_stream_t s) {
    PULSE_REQUIRE(m == 0 || b != nullptr, "pulse_quat_mul: null b");
    return launch<OP_QMUL>("pulse_quat_mul", a, b, nullptr, out, nullptr, m, s);
}
int pulse_quat_conjugate(const float* a, float* out, int64_t m, pulse_stream_t s) {
    return launch<OP_QCONJ>("pulse_quat_conjugate", a, nullptr, nullptr, out, nullptr, m, s);
}
int pulse_quat_rotate(const float* q, const float* v, float* out, int64_t m, pulse_stream_t s) {
    PULSE_REQUIRE(m == 0 || v != nullptr, "pulse_quat_rotate: null v");
    return launch<OP_QROT>("pulse_quat_rotate", q, v, nullptr, out, nullptr, m, s);
}
int pulse_quat_to_angle_axis(const float* q, float* angle, float* axis, int64_t m, pulse_stream_t s) {
    PULSE_REQUIRE(m == 0 || axis != nullptr, "pulse_quat_to_angle_axis: null axis");
    return launch<OP_Q2AA>("pulse_quat_to_angle_axis", q, nullptr, nullptr, angle, axis, m, s);
}
int pulse_quat_to_exp_map(const float* q, float* out, int64_t m, pulse_stream_t s) {
    return launch<OP_Q2EXP>("pulse_quat_to_exp_map", q, nullptr, nullptr, out, nullptr, m, s);
}
int pulse_quat_to_tan_norm(const float* q, float* out, int64_t m, pulse_stream_t s) {
    return launch<OP_Q2TN>("pulse_quat_to_tan_norm", q, nullptr, nullptr, out, nullptr, m, s);
}
int pulse_exp_map_to_quat(const float* e, float* out, int64_t m, pulse_stream_t s) {
    return launch<OP_EXP2Q>("pulse_exp_map_to_quat", e, nullptr, nullptr, out, nullptr, m, s);
}
int pulse_slerp(const float* q0, const float* q1, const float* t, float* out, int64_t m, pulse_stream_t s) {
    PULSE_REQUIRE(m == 0 || (q1 != nullptr && t != nullptr), "pulse_slerp: null input");
    return launch<OP_SLERP>("pulse_slerp", q0, q1, t, out, nullptr, m, s);
}
int pulse_calc_heading(const float* q, float* out, int64_t m, pulse_stream_t s) {
    return launch<OP_HEADING>("pulse_calc_heading", q, nullptr, nullptr, out, nullptr, m, s);
}
int pulse_calc_heading_quat(const float* q, float* out, int64_t m, int inverse, pulse_stream_t s) {
    return inverse ? launch<OP_HEADQ_INV>("pulse_calc_heading_quat", q, nullptr, nullptr, out, nullptr, m, s)
                   : launch<OP_HEADQ>("pulse_calc_heading_quat", q, nullptr, nullptr, out, nullptr, m, s);
}
}
