// Reference-motion query for gfx950: MotionLibBase.get_motion_state / get_root_pos_smpl / _calc_frame_blend
// (phc/utils/motion_lib_base.py:434-565).
//
// Per query (motion id, time): frame pair + blend from the clip's length / frame count / dt, then for every body
// lerp of position (+ per-env offset), velocity and angular velocity, slerp of the global rotation, and for every
// dof joint slerp of the LOCAL rotation -> exp-map (dof_pos) and lerp of the dof velocity.
//
// Mapping: a 32-lane half-wave per query (8 queries per 256-thread block); lane j <-> body j and dof joint j (= body
// j+1).  The two frame records are contiguous 1920-B rows of the packed table (include/pulse_hip.h section 2b), so the
// per-field reads of adjacent lanes are adjacent addresses: the gather costs whole cache lines of two rows instead of
// twelve scattered table rows.  Irregular-gather / HBM bound: 2 x 1908 B read, <= 1.85 KB written per query.
// Compiled with -ffp-contract=off (reference operation order: (1-b)*x0 + b*x1 [+ offset]).
#include "common.h"
#include "rot_math.h"

namespace pulse {

constexpr int kMsLanes = 32;
constexpr int kMsQueries = 8;

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

__device__ __forceinline__ float query_time(const pulse_motion_state_args& a, long long i, long long e) {
    if (a.motion_times) return a.motion_times[i];
    float t = (float)(a.progress[e] + a.step_shift) * a.dt;
    if (a.time_steps > 1) t = t + (float)(i - e * a.time_steps) * a.traj_dt;
    if (a.start_times) t = t + a.start_times[e];
    if (a.start_offsets) t = t + a.start_offsets[e];
    return t;
}

__device__ __forceinline__ V3 lerp3(const float* p0, const float* p1, float b) {
    const float a = 1.0f - b;
    return V3{a * p0[0] + b * p1[0], a * p0[1] + b * p1[1], a * p0[2] + b * p1[2]};
}

__global__ void __launch_bounds__(kMsQueries * kMsLanes) motion_state_kernel(const pulse_motion_state_args a) {
    const int slot = threadIdx.x / kMsLanes, lane = threadIdx.x % kMsLanes;
    const long long i = (long long)blockIdx.x * kMsQueries + slot;
    if (i >= a.n) return;
    const pulse_motion_tables& T = a.tab;
    const long long e = a.motion_times ? i : i / (a.time_steps > 1 ? a.time_steps : 1);     // per-env arrays in clock mode
    const long long m = a.motion_ids[e];
    const FrameBlend fb = calc_frame_blend(query_time(a, i, e), T.motion_lengths[m], T.motion_num_frames[m], T.motion_dt[m]);
    if (lane == 0) {
        if (a.frame_idx0) a.frame_idx0[i] = fb.f0;
        if (a.frame_idx1) a.frame_idx1[i] = fb.f1;
        if (a.blend) a.blend[i] = fb.blend;
    }
    const long long base = T.length_starts[m];
    const float* r0 = T.frames + (base + fb.f0) * T.frame_stride;
    const float* r1 = T.frames + (base + fb.f1) * T.frame_stride;
    const float b = fb.blend;
    const int J = T.num_bodies;
    if (a.root_only) {
        if (lane == 0 && a.root_pos) {
            const V3 p = lerp3(r0 + T.off_gts, r1 + T.off_gts, b);
            a.root_pos[3 * i] = p.x; a.root_pos[3 * i + 1] = p.y; a.root_pos[3 * i + 2] = p.z;
        }
        return;
    }
    if (lane < J) {
        const int j = lane;
        float* rec = a.rb_records ? a.rb_records + i * a.rb_query_stride + 13 * j : nullptr;
        if (a.rg_pos || rec) {
            V3 p = lerp3(r0 + T.off_gts + 3 * j, r1 + T.off_gts + 3 * j, b);
            if (a.offset) { p.x = p.x + a.offset[3 * e]; p.y = p.y + a.offset[3 * e + 1]; p.z = p.z + a.offset[3 * e + 2]; }
            if (a.rg_pos) { float* o = a.rg_pos + (i * J + j) * 3; o[0] = p.x; o[1] = p.y; o[2] = p.z; }
            if (rec) { rec[0] = p.x; rec[1] = p.y; rec[2] = p.z; }
        }
        if (a.body_vel || rec) {
            const V3 v = lerp3(r0 + T.off_gvs + 3 * j, r1 + T.off_gvs + 3 * j, b);
            if (a.body_vel) { float* o = a.body_vel + (i * J + j) * 3; o[0] = v.x; o[1] = v.y; o[2] = v.z; }
            if (rec) { rec[7] = v.x; rec[8] = v.y; rec[9] = v.z; }
        }
        if (a.body_ang_vel || rec) {
            const V3 w = lerp3(r0 + T.off_gavs + 3 * j, r1 + T.off_gavs + 3 * j, b);
            if (a.body_ang_vel) { float* o = a.body_ang_vel + (i * J + j) * 3; o[0] = w.x; o[1] = w.y; o[2] = w.z; }
            if (rec) { rec[10] = w.x; rec[11] = w.y; rec[12] = w.z; }
        }
        if (a.rb_rot || rec) {
            const float4 q0 = *reinterpret_cast<const float4*>(r0 + T.off_grs + 4 * j);
            const float4 q1 = *reinterpret_cast<const float4*>(r1 + T.off_grs + 4 * j);
            const Q4 q = slerp(Q4{q0.x, q0.y, q0.z, q0.w}, Q4{q1.x, q1.y, q1.z, q1.w}, b);
            if (a.rb_rot) *reinterpret_cast<float4*>(a.rb_rot + (i * J + j) * 4) = make_float4(q.x, q.y, q.z, q.w);
            if (rec) { rec[3] = q.x; rec[4] = q.y; rec[5] = q.z; rec[6] = q.w; }
        }
    }
    if (lane < J - 1) {
        const int d = lane;                                          // dof joint d <-> body d + 1
        const int nd = 3 * (J - 1);
        if (a.dof_pos) {
            const float4 q0 = *reinterpret_cast<const float4*>(r0 + T.off_lrs + 4 * (d + 1));
            const float4 q1 = *reinterpret_cast<const float4*>(r1 + T.off_lrs + 4 * (d + 1));
            const Q4 q = slerp(Q4{q0.x, q0.y, q0.z, q0.w}, Q4{q1.x, q1.y, q1.z, q1.w}, b);
            V3 ax;
            const float ang = q_to_angle_axis(q, &ax);               // quat_to_exp_map, torch_utils.py:81-97
            float* o = a.dof_pos + i * nd + 3 * d;
            o[0] = ang * ax.x; o[1] = ang * ax.y; o[2] = ang * ax.z;
        }
        if (a.dof_vel) {
            const V3 v = lerp3(r0 + T.off_dvs + 3 * d, r1 + T.off_dvs + 3 * d, b);
            float* o = a.dof_vel + i * nd + 3 * d;
            o[0] = v.x; o[1] = v.y; o[2] = v.z;
        }
    }
}

}  // namespace pulse

extern "C" int pulse_sizeof_motion_state_args(void) { return (int)sizeof(pulse_motion_state_args); }

extern "C" int pulse_motion_state(const pulse_motion_state_args* args, pulse_stream_t s) {
    using namespace pulse;
    PULSE_REQUIRE(args != nullptr, "pulse_motion_state: null args");
    const pulse_motion_state_args& a = *args;
    const pulse_motion_tables& T = a.tab;
    PULSE_REQUIRE(a.n >= 0, "pulse_motion_state: negative query count");
    if (a.n == 0) return PULSE_OK;
    PULSE_REQUIRE(T.frames && T.motion_lengths && T.motion_dt && T.motion_num_frames && T.length_starts, "pulse_motion_state: null table pointer");
    PULSE_REQUIRE(T.num_bodies >= 1 && T.num_bodies <= kMsLanes, "pulse_motion_state: num_bodies must be in [1, 32]");
    PULSE_REQUIRE(T.frame_stride % 4 == 0 && T.off_grs % 4 == 0 && T.off_lrs % 4 == 0, "pulse_motion_state: record pitch / quaternion fields must be 16-B aligned");
    PULSE_REQUIRE(T.off_gts >= 0 && T.off_grs >= 0 && T.off_lrs >= 0 && T.off_gvs >= 0 && T.off_gavs >= 0 && T.off_dvs >= 0, "pulse_motion_state: negative field offset");
    PULSE_REQUIRE(a.motion_ids != nullptr, "pulse_motion_state: null motion_ids");
    PULSE_REQUIRE(a.motion_times != nullptr || a.progress != nullptr, "pulse_motion_state: need motion_times or progress");
    PULSE_REQUIRE(!a.root_only || a.root_pos != nullptr, "pulse_motion_state: root_only needs root_pos");
    PULSE_REQUIRE(a.motion_times != nullptr || a.time_steps <= 1 || a.n % a.time_steps == 0, "pulse_motion_state: n must be num_envs * time_steps");
    PULSE_REQUIRE(a.rb_records == nullptr || a.rb_query_stride >= 13 * T.num_bodies, "pulse_motion_state: rb_query_stride too small");
    const long long blocks = (a.n + kMsQueries - 1) / kMsQueries;
    hipLaunchKernelGGL(motion_state_kernel, dim3((unsigned)blocks), dim3(kMsQueries * kMsLanes), 0, as_stream(s), a);
    return check_launch("pulse_motion_state");
}
