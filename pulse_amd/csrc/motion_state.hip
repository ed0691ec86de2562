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
#include "motion_math.h"

namespace pulse {

constexpr int kMsLanes = 32;
constexpr int kMsQueries = 8;

__device__ __forceinline__ float query_time(const pulse_motion_state_args& a, long long i, long long e) {
    if (a.motion_times) return a.motion_times[i];
    float t = (float)(a.progress[e] + a.step_shift) * a.dt;
    if (a.time_steps > 1) t = t + (float)(i - e * a.time_steps) * a.traj_dt;
    if (a.start_times) t = t + a.start_times[e];
    if (a.start_offsets) t = t + a.start_offsets[e];
    return t;
}

__global__ void __launch_bounds__(kMsQueries * kMsLanes) motion_state_kernel(const pulse_motion_state_args a) {
    const int slot = threadIdx.x / kMsLanes, lane = threadIdx.x % kMsLanes;
    const long long i = (long long)blockIdx.x * kMsQueries + slot;
    if (i >= a.n) return;
    const pulse_motion_tables& T = a.tab;
    const long long e = a.motion_times ? i : i / (a.time_steps > 1 ? a.time_steps : 1);     // per-env arrays in clock mode
    const long long m = a.motion_ids[e];
    float t;
    if (a.reset_mask) {
        // reference-state init of the masked envs (_reset_envs -> _sample_ref_state, humanoid_im.py:966-986): new start time
        // phase * motion length (sample_time, motion_lib_base.py:401-411), episode clock back to 0, state := reference at that time
        if (a.reset_mask[e] == 0) return;
        float st = a.reset_phase ? a.reset_phase[e] * T.motion_lengths[m] : 0.0f;
        if (a.reset_phase && a.reset_time_interval) {
            const float curr_fps = (float)(1.0 / 30.0);              // sample_time_interval: ((phase * len) / curr_fps).long() * curr_fps
            st = (float)((long long)(st / curr_fps)) * curr_fps;
        }
        t = (float)a.step_shift * a.dt + st;
        if (a.start_offsets && !a.reset_start_offsets) t = t + a.start_offsets[e];
        if (lane == 0) {
            if (a.reset_start_times) a.reset_start_times[e] = st;
            if (a.reset_progress) a.reset_progress[e] = 0;
            if (a.reset_clear0) a.reset_clear0[e] = 0;
            if (a.reset_clear1) a.reset_clear1[e] = 0;
            if (a.reset_clear2) a.reset_clear2[e] = 0;
            if (a.reset_start_offsets) a.reset_start_offsets[e] = 0.0f;
            if (a.reset_global_offset) { a.reset_global_offset[3 * e] = 0.0f; a.reset_global_offset[3 * e + 1] = 0.0f; a.reset_global_offset[3 * e + 2] = 0.0f; }
        }
    } else {
        t = query_time(a, i, e);
    }
    const FramePair fp = frame_pair(T, m, t);
    if (lane == 0) {
        if (a.frame_idx0) a.frame_idx0[i] = fp.f0;
        if (a.frame_idx1) a.frame_idx1[i] = fp.f1;
        if (a.blend) a.blend[i] = fp.blend;
    }
    const int J = T.num_bodies;
    const float* off = (a.offset && !(a.reset_mask && a.reset_global_offset)) ? a.offset + 3 * e : nullptr;
    if (a.root_only) {
        if (lane == 0 && a.root_pos) {
            const V3 p = lerp3(fp.r0 + T.off_gts, fp.r1 + T.off_gts, fp.blend);
            a.root_pos[3 * i] = p.x; a.root_pos[3 * i + 1] = p.y; a.root_pos[3 * i + 2] = p.z;
        }
        return;
    }
    if (lane < J && (a.rg_pos || a.rb_rot || a.body_vel || a.body_ang_vel || a.rb_records)) {
        const int j = lane;
        const BodyState s = blend_body(T, fp.r0, fp.r1, fp.blend, j, off);
        if (a.rg_pos) { float* o = a.rg_pos + (i * J + j) * 3; o[0] = s.p.x; o[1] = s.p.y; o[2] = s.p.z; }
        if (a.body_vel) { float* o = a.body_vel + (i * J + j) * 3; o[0] = s.v.x; o[1] = s.v.y; o[2] = s.v.z; }
        if (a.body_ang_vel) { float* o = a.body_ang_vel + (i * J + j) * 3; o[0] = s.w.x; o[1] = s.w.y; o[2] = s.w.z; }
        if (a.rb_rot) *reinterpret_cast<float4*>(a.rb_rot + (i * J + j) * 4) = make_float4(s.q.x, s.q.y, s.q.z, s.q.w);
        if (a.rb_records) {
            float* rec = a.rb_records + i * a.rb_query_stride + 13 * j;
            rec[0] = s.p.x; rec[1] = s.p.y; rec[2] = s.p.z;
            rec[3] = s.q.x; rec[4] = s.q.y; rec[5] = s.q.z; rec[6] = s.q.w;
            rec[7] = s.v.x; rec[8] = s.v.y; rec[9] = s.v.z;
            rec[10] = s.w.x; rec[11] = s.w.y; rec[12] = s.w.z;
        }
    }
    if (lane < J - 1 && (a.dof_pos || a.dof_vel)) {
        const int d = lane, nd = 3 * (J - 1);
        V3 dp, dv;
        blend_dof(T, fp.r0, fp.r1, fp.blend, d, &dp, &dv);
        if (a.dof_pos) { float* o = a.dof_pos + i * nd + 3 * d; o[0] = dp.x; o[1] = dp.y; o[2] = dp.z; }
        if (a.dof_vel) { float* o = a.dof_vel + i * nd + 3 * d; o[0] = dv.x; o[1] = dv.y; o[2] = dv.z; }
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
    PULSE_REQUIRE(a.motion_times != nullptr || a.progress != nullptr || a.reset_mask != nullptr, "pulse_motion_state: need motion_times, progress or reset_mask");
    PULSE_REQUIRE(a.reset_mask == nullptr || (a.motion_times == nullptr && a.time_steps <= 1), "pulse_motion_state: reset mode is per env");
    PULSE_REQUIRE(!a.root_only || a.root_pos != nullptr, "pulse_motion_state: root_only needs root_pos");
    PULSE_REQUIRE(a.motion_times != nullptr || a.time_steps <= 1 || a.n % a.time_steps == 0, "pulse_motion_state: n must be num_envs * time_steps");
    PULSE_REQUIRE(a.rb_records == nullptr || a.rb_query_stride >= 13 * T.num_bodies, "pulse_motion_state: rb_query_stride too small");
    const long long blocks = (a.n + kMsQueries - 1) / kMsQueries;
    hipLaunchKernelGGL(motion_state_kernel, dim3((unsigned)blocks), dim3(kMsQueries * kMsLanes), 0, as_stream(s), a);
    return check_launch("pulse_motion_state");
}
