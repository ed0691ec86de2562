// Device-side quaternion / exp-map / 6-D rotation math (xyzw, fp32).
//
// Operation order follows the reference's TorchScript functions so that results
// track the PyTorch-CPU oracle to round-off; the translation units that include
// this header are compiled with -ffp-contract=off (no FMA fusion) for the same
// reason (hard part 2: theta = 2 acos(w) amplifies any
// difference in w when the tracking error is small).
#pragma once
#include <hip/hip_runtime.h>

namespace pulse {

struct Q4 { float x, y, z, w; };
struct V3 { float x, y, z; };

// isaacgym.torch_utils.quat_mul, factored 8-multiply form.
__device__ __forceinline__ Q4 qmul(const Q4 a, const Q4 b) {
    const float ww = (a.z + a.x) * (b.x + b.y);
    const float yy = (a.w - a.y) * (b.w + b.z);
    const float zz = (a.w + a.y) * (b.w - b.z);
    const float xx = ww + yy + zz;
    const float qq = 0.5f * (xx + (a.z - a.x) * (b.x - b.y));
    Q4 r;
    r.w = qq - ww + (a.z - a.y) * (b.y - b.z);
    r.x = qq - xx + (a.x + a.w) * (b.x + b.w);
    r.y = qq - yy + (a.w - a.x) * (b.y + b.z);
    r.z = qq - zz + (a.z + a.y) * (b.w - b.x);
    return r;
}

__device__ __forceinline__ Q4 qconj(const Q4 a) { return Q4{-a.x, -a.y, -a.z, a.w}; }

// my_quat_rotate, phc/utils/torch_utils.py:45-55:  v(2w^2-1) + 2w(u x v) + 2u(u.v)
__device__ __forceinline__ V3 qrot(const Q4 q, const V3 v) {
    const float s = 2.0f * (q.w * q.w) - 1.0f;
    const float cx = q.y * v.z - q.z * v.y;
    const float cy = q.z * v.x - q.x * v.z;
    const float cz = q.x * v.y - q.y * v.x;
    const float d = q.x * v.x + q.y * v.y + q.z * v.z;
    V3 r;
    r.x = v.x * s + cx * q.w * 2.0f + q.x * d * 2.0f;
    r.y = v.y * s + cy * q.w * 2.0f + q.y * d * 2.0f;
    r.z = v.z * s + cz * q.w * 2.0f + q.z * d * 2.0f;
    return r;
}

// isaacgym normalize_angle: atan2(sin x, cos x)
__device__ __forceinline__ float wrap_angle(float x) { return atan2f(sinf(x), cosf(x)); }

// isaacgym normalize(x, eps=1e-9) for 3- and 4-vectors
__device__ __forceinline__ V3 unit3(const V3 v) {
    const float n = fmaxf(sqrtf(v.x * v.x + v.y * v.y + v.z * v.z), 1e-9f);
    return V3{v.x / n, v.y / n, v.z / n};
}
__device__ __forceinline__ Q4 unit4(const Q4 q) {
    const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-9f);
    return Q4{q.x / n, q.y / n, q.z / n, q.w / n};
}

// isaacgym quat_from_angle_axis
__device__ __forceinline__ Q4 q_from_angle_axis(float angle, const V3 axis) {
    const float h = angle / 2.0f;
    const V3 a = unit3(axis);
    const float sh = sinf(h);
    return unit4(Q4{a.x * sh, a.y * sh, a.z * sh, cosf(h)});
}

// quat_to_angle_axis, phc/utils/torch_utils.py:57-78.  NaNs (|w|>1) fall in the masked branch.
__device__ __forceinline__ float q_to_angle_axis(const Q4 q, V3* axis) {
    const float st = sqrtf(1.0f - q.w * q.w);
    float ang = wrap_angle(2.0f * acosf(q.w));
    const bool ok = fabsf(st) > 1e-5f;  // false for NaN
    if (axis) {
        *axis = ok ? V3{q.x / st, q.y / st, q.z / st} : V3{0.0f, 0.0f, 1.0f};
    }
    return ok ? ang : 0.0f;
}

// my_quat_rotate by a HEADING quaternion h = (0, 0, z, w) (calc_heading_quat(_inv): a rotation about +z, x = y = +-0 exactly).  The terms of qrot
// that carry h.x / h.y are exact zeros there (0 * v = +-0, a + +-0 = a), so leaving them out gives the SAME fp32 values (at most the sign of a
// zero differs) in 15 instead of 35 operations; ``s`` = 2 h.w^2 - 1 is shared by every rotation of a lane.  [r6]
__device__ __forceinline__ V3 qrot_heading(const Q4 h, const float s, const V3 v) {
    V3 r;
    r.x = v.x * s + (-(h.z * v.y)) * h.w * 2.0f;      // cx = h.y v.z - h.z v.y = -(h.z v.y)
    r.y = v.y * s + (h.z * v.x) * h.w * 2.0f;         // cy = h.z v.x - h.x v.z =   h.z v.x
    r.z = v.z * s + h.z * (h.z * v.z) * 2.0f;         // cz = 0;  d = h.z v.z
    return r;
}

// quat_to_tan_norm, phc/utils/torch_utils.py:100-113: [R x^, R z^] = my_quat_rotate(q, (1, 0, 0)), my_quat_rotate(q, (0, 0, 1)).
// Written out for the two unit vectors [r6]: the products with their zero components are exact zeros and 1 * x = x, so the values are those of
// the two general rotations (at most the sign of a zero differs) in 26 instead of 70 operations.
__device__ __forceinline__ void q_to_tan_norm(const Q4 q, float out[6]) {
    const float s = 2.0f * (q.w * q.w) - 1.0f;
    // v = x^: cross = (0, q.z, -q.y), dot = q.x
    out[0] = s + q.x * q.x * 2.0f;
    out[1] = q.z * q.w * 2.0f + q.y * q.x * 2.0f;
    out[2] = (-q.y) * q.w * 2.0f + q.z * q.x * 2.0f;
    // v = z^: cross = (q.y, -q.x, 0), dot = q.z
    out[3] = q.y * q.w * 2.0f + q.x * q.z * 2.0f;
    out[4] = (-q.x) * q.w * 2.0f + q.y * q.z * 2.0f;
    out[5] = s + q.z * q.z * 2.0f;
}
// the two general rotations q_to_tan_norm stands for (tests compare the two forms)
__device__ __forceinline__ void q_to_tan_norm_general(const Q4 q, float out[6]) {
    const V3 t = qrot(q, V3{1.0f, 0.0f, 0.0f});
    const V3 n = qrot(q, V3{0.0f, 0.0f, 1.0f});
    out[0] = t.x; out[1] = t.y; out[2] = t.z;
    out[3] = n.x; out[4] = n.y; out[5] = n.z;
}

// calc_heading, phc/utils/torch_utils.py:200-212
__device__ __forceinline__ float heading_angle(const Q4 q) {
    // my_quat_rotate(q, x^).xy, written out as in q_to_tan_norm (same values as the general rotation) [r6]
    const float s = 2.0f * (q.w * q.w) - 1.0f;
    const float dx = s + q.x * q.x * 2.0f;
    const float dy = q.z * q.w * 2.0f + q.y * q.x * 2.0f;
    return atan2f(dy, dx);
}
// calc_heading_quat(_inv), phc/utils/torch_utils.py:215-240
__device__ __forceinline__ Q4 heading_quat(const Q4 q, bool inverse) {
    const float h = heading_angle(q);
    return q_from_angle_axis(inverse ? -h : h, V3{0.0f, 0.0f, 1.0f});
}

// exp_map_to_angle_axis + exp_map_to_quat, phc/utils/torch_utils.py:148-172
__device__ __forceinline__ Q4 exp_map_to_q(const V3 e) {
    const float n = sqrtf(e.x * e.x + e.y * e.y + e.z * e.z);
    V3 ax{e.x / n, e.y / n, e.z / n};
    float ang = wrap_angle(n);
    const bool ok = fabsf(ang) > 1e-5f;
    if (!ok) { ang = 0.0f; ax = V3{0.0f, 0.0f, 1.0f}; }
    return q_from_angle_axis(ang, ax);
}

// slerp, phc/utils/torch_utils.py:175-197
__device__ __forceinline__ Q4 slerp(const Q4 q0, Q4 q1, float t) {
    float c = q0.x * q1.x + q0.y * q1.y + q0.z * q1.z + q0.w * q1.w;
    if (c < 0.0f) { q1 = Q4{-q1.x, -q1.y, -q1.z, -q1.w}; }
    c = fabsf(c);
    const float half = acosf(c);
    const float sh = sqrtf(1.0f - c * c);
    const float ra = sinf((1.0f - t) * half) / sh;
    const float rb = sinf(t * half) / sh;
    Q4 r{ra * q0.x + rb * q1.x, ra * q0.y + rb * q1.y, ra * q0.z + rb * q1.z, ra * q0.w + rb * q1.w};
    if (fabsf(sh) < 0.001f) {
        r = Q4{0.5f * q0.x + 0.5f * q1.x, 0.5f * q0.y + 0.5f * q1.y, 0.5f * q0.z + 0.5f * q1.z, 0.5f * q0.w + 0.5f * q1.w};
    }
    if (fabsf(c) >= 1.0f) r = q0;
    return r;
}

}  // namespace pulse
