// Device math shared by the reference-motion kernels (motion_state.hip, env_step.hip):
// MotionLibBase._calc_frame_blend and the per-body blend of two packed frame records
// (phc/utils/motion_lib_base.py:434-517, 546-557).  Compile with -ffp-contract=off.
#pragma once
#include "common.h"
#include "rot_math.h"

namespace pulse {

struct FrameBlend { long long f0, f1; float blend; };

// _calc_frame_blend, motion_lib_base.py:546-557
__device__ __forceinline__ FrameBlend calc_frame_blend(float time, float len, long long num_frames, float dt) {
    float phase = time / len;
    phase = fminf(fmaxf(phase, 0.0f), 1.0f);
    if (time != time) phase = time;                              // torch.clip propagates NaN
    if (time < 0.0f) time = 0.0f;
    FrameBlend r;
    r.f0 = (long long)(phase * (float)(num_frames - 1));
    if (!(r.f0 >= 0)) r.f0 = 0;                                  // NaN time: stay inside the clip (the reference would raise)
    if (r.f0 > num_frames - 1) r.f0 = num_frames - 1;
    r.f1 = r.f0 + 1 < num_frames - 1 ? r.f0 + 1 : num_frames - 1;
    const float b = (time - (float)r.f0 * dt) / dt;
    r.blend = (b != b) ? b : fminf(fmaxf(b, 0.0f), 1.0f);
    return r;
}

__device__ __forceinline__ V3 lerp3(const float* p0, const float* p1, float b) {
    const float a = 1.0f - b;
    return V3{a * p0[0] + b * p1[0], a * p0[1] + b * p1[1], a * p0[2] + b * p1[2]};
}

// body j of the blended frame: position (+ offset), global rotation, linear / angular velocity
struct BodyState { V3 p; Q4 q; V3 v; V3 w; };

__device__ __forceinline__ BodyState blend_body(const pulse_motion_tables& T, const float* r0, const float* r1, float b, int j, const float* offset) {
    BodyState s;
    s.p = lerp3(r0 + T.off_gts + 3 * j, r1 + T.off_gts + 3 * j, b);
    if (offset) { s.p.x = s.p.x + offset[0]; s.p.y = s.p.y + offset[1]; s.p.z = s.p.z + offset[2]; }
    s.v = lerp3(r0 + T.off_gvs + 3 * j, r1 + T.off_gvs + 3 * j, b);
    s.w = lerp3(r0 + T.off_gavs + 3 * j, r1 + T.off_gavs + 3 * j, b);
    const float4 q0 = *reinterpret_cast<const float4*>(r0 + T.off_grs + 4 * j);
    const float4 q1 = *reinterpret_cast<const float4*>(r1 + T.off_grs + 4 * j);
    s.q = slerp(Q4{q0.x, q0.y, q0.z, q0.w}, Q4{q1.x, q1.y, q1.z, q1.w}, b);
    return s;
}

// dof joint d (= body d + 1): exp-map of the slerped local rotation and the blended dof velocity
__device__ __forceinline__ void blend_dof(const pulse_motion_tables& T, const float* r0, const float* r1, float b, int d, V3* pos, V3* vel) {
    const float4 q0 = *reinterpret_cast<const float4*>(r0 + T.off_lrs + 4 * (d + 1));
    const float4 q1 = *reinterpret_cast<const float4*>(r1 + T.off_lrs + 4 * (d + 1));
    const Q4 q = slerp(Q4{q0.x, q0.y, q0.z, q0.w}, Q4{q1.x, q1.y, q1.z, q1.w}, b);
    V3 ax;
    const float ang = q_to_angle_axis(q, &ax);               // quat_to_exp_map, torch_utils.py:81-97
    *pos = V3{ang * ax.x, ang * ax.y, ang * ax.z};
    *vel = lerp3(r0 + T.off_dvs + 3 * d, r1 + T.off_dvs + 3 * d, b);
}

// the two frame records of (motion m, time t)
struct FramePair { const float* r0; const float* r1; float blend; long long f0, f1; };

__device__ __forceinline__ FramePair frame_pair(const pulse_motion_tables& T, long long m, float t) {
    const FrameBlend fb = calc_frame_blend(t, T.motion_lengths[m], T.motion_num_frames[m], T.motion_dt[m]);
    const long long base = T.length_starts[m];
    return FramePair{T.frames + (base + fb.f0) * T.frame_stride, T.frames + (base + fb.f1) * T.frame_stride, fb.blend, fb.f0, fb.f1};
}

}  // namespace pulse
