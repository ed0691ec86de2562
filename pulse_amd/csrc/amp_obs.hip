// AMP (adversarial motion prior) per-frame observation for gfx950.
//
// Replaces build_amp_observations_smpl (phc/env/tasks/humanoid_amp.py:925-969) and dof_to_obs_smpl
// (phc/env/tasks/humanoid.py:1436-1446): per env
//   [root_h (1)?, 6-D heading-local root rot, local root lin vel (3), local root ang vel (3),
//    6-D rotation of every (selected) dof joint from its exp-map (6 Jd), dof velocities (3 Jd),
//    heading-local key-body positions (3 Kb)]
// = 232 floats for the 23 SMPL joints / 4 key bodies (196 for the 19-joint subset).
//
// Same mapping as the fused env step: a 32-lane half-wave per env, lane j <-> joint j / key body j, inputs
// read straight from the 13-float rigid-body records and the (N, 69) dof tensors, the feature row assembled
// in LDS and written with coalesced stores into the caller's pitch (e.g. slot 0 of the (N, 10, W) history).
// HBM/launch bound: 52 B root record + 2 x 276 B dofs + 48 B key bodies read, 4 W bytes written per env.
// Compiled with -ffp-contract=off (reference operation order).
#include "common.h"
#include "rot_math.h"
#include "motion_math.h"

namespace pulse {

constexpr int kAmpLanes = 32;
constexpr int kAmpEnvs = 4;
constexpr int kAmpMaxW = 320;

constexpr int kAmpMaxHist = 9;      // history frames behind the current one (numAMPObsSteps - 1)

// HIST: ``out`` is slot 0 of the env's (hist_steps, W) history window (HumanoidAMP._amp_obs_buf, humanoid_amp.py:296-314).  The launch
// then does the whole per-step update of HumanoidAMP.post_physics_step (:194-210) for the env: _update_hist_amp_obs (frames 0 .. S-2 move
// to 1 .. S-1, :622-631), the current frame into slot 0, and -- window_out -- the finished window copied to the caller's row (the
// experience-buffer slot of this rollout step: amp_agent.py:377).  Four device copies of the (N, S W) window become none.
template <bool HIST>
__global__ void __launch_bounds__(kAmpEnvs * kAmpLanes) amp_obs_kernel(const pulse_amp_obs_args a) {
    __shared__ float s_out[kAmpEnvs][kAmpMaxW];
    __shared__ float s_hist[HIST ? kAmpEnvs : 1][HIST ? kAmpMaxHist * kAmpMaxW : 1];
    const int slot = threadIdx.x / kAmpLanes, lane = threadIdx.x % kAmpLanes;
    const int idx = blockIdx.x * kAmpEnvs + slot;
    const int count = a.env_ids ? a.num_ids : a.num_envs;
    bool valid = idx < count;
    long long e = 0;
    if (valid) {
        e = a.env_ids ? a.env_ids[idx] : (long long)idx;
        if (a.env_mask && a.env_mask[e] == 0) valid = false;
    }
    const int Jd = a.num_joints, Kb = a.num_key_bodies;
    const int h0 = a.root_height_obs ? 1 : 0;
    const int off_vel = h0 + 6, off_ang = off_vel + 3, off_dof = off_ang + 3, off_dvel = off_dof + 6 * Jd, off_key = off_dvel + 3 * Jd;
    const int W = off_key + 3 * Kb;
    float* o = s_out[slot];
    if constexpr (HIST) {
        if (valid) {                                                     // the frames that move down one slot (read before anything is written)
            const float* h = a.out + e * a.out_stride;
            const int n4 = (a.hist_steps - 1) * W / 4;                   // W and the window base are multiples of 4 floats (checked by the launcher)
            for (int c = lane; c < n4; c += kAmpLanes) reinterpret_cast<float4*>(s_hist[slot])[c] = reinterpret_cast<const float4*>(h)[c];
        }
    }
    if (valid) {
        const float* rb = a.rb + e * a.rb_env_stride;
        const V3 root_p{rb[0], rb[1], rb[2]};
        const Q4 root_q{rb[3], rb[4], rb[5], rb[6]};
        const Q4 hinv = heading_quat(root_q, true);
        if (lane == 0) {
            if (a.root_height_obs) o[0] = root_p.z;
            float tn[6];
            q_to_tan_norm(a.local_root_obs ? qmul(hinv, root_q) : root_q, tn);
#pragma unroll
            for (int k = 0; k < 6; ++k) o[h0 + k] = tn[k];
            const V3 lv = qrot(hinv, V3{rb[7], rb[8], rb[9]});
            const V3 lw = qrot(hinv, V3{rb[10], rb[11], rb[12]});
            o[off_vel] = lv.x; o[off_vel + 1] = lv.y; o[off_vel + 2] = lv.z;
            o[off_ang] = lw.x; o[off_ang + 1] = lw.y; o[off_ang + 2] = lw.z;
        }
        if (lane < Jd) {
            const int j = a.joint_ids ? a.joint_ids[lane] : lane;
            const bool zeroed = a.zero_joint_mask & (1u << j);     // the "ZL hack" (humanoid_amp.py:636-639): toes / hands read as 0
            const float* dp = a.dof_pos + e * a.num_dof + 3 * j;
            const float* dv = a.dof_vel + e * a.num_dof + 3 * j;
            const V3 em = zeroed ? V3{0.f, 0.f, 0.f} : V3{dp[0], dp[1], dp[2]};
            float tn[6];
            q_to_tan_norm(exp_map_to_q(em), tn);
#pragma unroll
            for (int k = 0; k < 6; ++k) o[off_dof + 6 * lane + k] = tn[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) o[off_dvel + 3 * lane + k] = zeroed ? 0.f : dv[k];
        }
        if (lane < Kb) {
            const float* kb = rb + 13 * a.key_body_ids[lane];
            const V3 lp = qrot(hinv, V3{kb[0] - root_p.x, kb[1] - root_p.y, kb[2] - root_p.z});
            o[off_key + 3 * lane] = lp.x; o[off_key + 3 * lane + 1] = lp.y; o[off_key + 3 * lane + 2] = lp.z;
        }
    }
    __syncthreads();
    if (valid) {
        float* g = a.out + e * a.out_stride;
        for (int c = lane; c < W; c += kAmpLanes) g[c] = o[c];
        if constexpr (HIST) {
            const int n4 = (a.hist_steps - 1) * W / 4, w4 = W / 4;
            float4* gh = reinterpret_cast<float4*>(g + W);
            const float4* sh = reinterpret_cast<const float4*>(s_hist[slot]);
            for (int c = lane; c < n4; c += kAmpLanes) gh[c] = sh[c];
            if (a.window_out) {
                float* wo = a.window_out + e * a.window_stride;
                for (int c = lane; c < w4; c += kAmpLanes) reinterpret_cast<float4*>(wo)[c] = reinterpret_cast<const float4*>(o)[c];
                float4* wh = reinterpret_cast<float4*>(wo + W);
                for (int c = lane; c < n4; c += kAmpLanes) wh[c] = sh[c];
            }
        }
    }
}

// _init_amp_obs_ref (humanoid_amp.py:531-563) for the masked envs in ONE launch: history slot k + 1 of env e := the AMP frame of the
// env's motion at start_time[e] - dt (k + 1) -- MotionLib.get_motion_state (motion_lib_base.py:434-517) and build_amp_observations_smpl
// fused, a half-wave per (env, k); envs outside the mask cost an index read.  (Was: a motion query and an AMP-frame pass over ALL
// N x (S - 1) rows plus a where / copy of the window, 195 us per rollout step at 8192 envs.)
__global__ void __launch_bounds__(kAmpEnvs * kAmpLanes) amp_hist_init_kernel(const pulse_amp_hist_args a) {
    __shared__ float s_out[kAmpEnvs][kAmpMaxW];
    const int slot = threadIdx.x / kAmpLanes, lane = threadIdx.x % kAmpLanes;
    const long long idx = (long long)blockIdx.x * kAmpEnvs + slot;
    const int Hs = a.hist_steps - 1;
    bool valid = idx < (long long)a.num_envs * Hs;
    const long long e = valid ? idx / Hs : 0;
    const int k = valid ? (int)(idx - e * Hs) : 0;
    if (valid && a.env_mask && a.env_mask[e] == 0) valid = false;
    const int Jd = a.num_joints, Kb = a.num_key_bodies;
    const int h0 = a.root_height_obs ? 1 : 0;
    const int off_vel = h0 + 6, off_ang = off_vel + 3, off_dof = off_ang + 3, off_dvel = off_dof + 6 * Jd, off_key = off_dvel + 3 * Jd;
    const int W = off_key + 3 * Kb;
    float* o = s_out[slot];
    if (valid) {
        const pulse_motion_tables& T = a.tab;
        // times = start_times[:, None] + (-dt * (arange(S - 1) + 1))   (fp32, the reference's op order)
        const float t = a.start_times[e] + (float)(k + 1) * (-a.dt);
        const FramePair fp = frame_pair(T, a.motion_ids[e], t);
        const BodyState root = blend_body(T, fp.r0, fp.r1, fp.blend, 0, nullptr);
        const Q4 hinv = heading_quat(root.q, true);
        if (lane == 0) {
            if (a.root_height_obs) o[0] = root.p.z;
            float tn[6];
            q_to_tan_norm(a.local_root_obs ? qmul(hinv, root.q) : root.q, tn);
#pragma unroll
            for (int c = 0; c < 6; ++c) o[h0 + c] = tn[c];
            const V3 lv = qrot(hinv, root.v), lw = qrot(hinv, root.w);
            o[off_vel] = lv.x; o[off_vel + 1] = lv.y; o[off_vel + 2] = lv.z;
            o[off_ang] = lw.x; o[off_ang + 1] = lw.y; o[off_ang + 2] = lw.z;
        }
        if (lane < Jd) {
            const int j = a.joint_ids ? a.joint_ids[lane] : lane;
            V3 dp, dv;
            blend_dof(T, fp.r0, fp.r1, fp.blend, j, &dp, &dv);
            float tn[6];
            q_to_tan_norm(exp_map_to_q(dp), tn);
#pragma unroll
            for (int c = 0; c < 6; ++c) o[off_dof + 6 * lane + c] = tn[c];
            o[off_dvel + 3 * lane] = dv.x; o[off_dvel + 3 * lane + 1] = dv.y; o[off_dvel + 3 * lane + 2] = dv.z;
        }
        if (lane < Kb) {
            const int b = a.key_body_ids[lane];
            const V3 kp = lerp3(fp.r0 + T.off_gts + 3 * b, fp.r1 + T.off_gts + 3 * b, fp.blend);
            const V3 lp = qrot(hinv, V3{kp.x - root.p.x, kp.y - root.p.y, kp.z - root.p.z});
            o[off_key + 3 * lane] = lp.x; o[off_key + 3 * lane + 1] = lp.y; o[off_key + 3 * lane + 2] = lp.z;
        }
    }
    __syncthreads();
    if (valid) {
        float* g = a.hist + e * a.env_stride + (long long)(k + 1) * a.step_stride;
        for (int c = lane; c < W; c += kAmpLanes) g[c] = o[c];
    }
}

}  // namespace pulse

using namespace pulse;

extern "C" {

int pulse_sizeof_amp_obs_args(void) { return (int)sizeof(pulse_amp_obs_args); }

int pulse_amp_obs_width(int num_joints, int num_key_bodies, int root_height_obs) {
    return (root_height_obs ? 1 : 0) + 12 + 9 * num_joints + 3 * num_key_bodies;
}

int pulse_amp_obs(const pulse_amp_obs_args* args, pulse_stream_t s) {
    PULSE_REQUIRE(args != nullptr, "pulse_amp_obs: null args");
    const pulse_amp_obs_args& a = *args;
    const int count = a.env_ids ? a.num_ids : a.num_envs;
    PULSE_REQUIRE(count >= 0, "pulse_amp_obs: negative count");
    if (count == 0) return PULSE_OK;
    PULSE_REQUIRE(a.rb && a.dof_pos && a.dof_vel && a.out && a.key_body_ids, "pulse_amp_obs: null pointer");
    PULSE_REQUIRE(a.num_joints >= 1 && a.num_joints <= kAmpLanes && a.num_key_bodies >= 0 && a.num_key_bodies <= kAmpLanes,
                  "pulse_amp_obs: joints / key bodies must fit 32 lanes");
    PULSE_REQUIRE(a.num_dof >= 3 * a.num_joints || a.joint_ids, "pulse_amp_obs: num_dof too small");
    const int w = pulse_amp_obs_width(a.num_joints, a.num_key_bodies, a.root_height_obs);
    PULSE_REQUIRE(w <= kAmpMaxW && a.out_stride >= w, "pulse_amp_obs: width %d exceeds %d or the output pitch", w, kAmpMaxW);
    if (a.hist_steps > 1) {
        PULSE_REQUIRE(a.hist_steps - 1 <= kAmpMaxHist && (w % 4) == 0 && (a.out_stride % 4) == 0 && a.out_stride >= (int64_t)a.hist_steps * w &&
                      (reinterpret_cast<uintptr_t>(a.out) & 15) == 0,
                      "pulse_amp_obs: history mode needs hist_steps <= %d, W and the window pitch multiples of 4 floats, a 16-byte aligned window", kAmpMaxHist + 1);
        PULSE_REQUIRE(!a.window_out || ((a.window_stride % 4) == 0 && a.window_stride >= (int64_t)a.hist_steps * w && (reinterpret_cast<uintptr_t>(a.window_out) & 15) == 0),
                      "pulse_amp_obs: window_out rows must be 16-byte aligned and hold hist_steps * W floats");
        hipLaunchKernelGGL(amp_obs_kernel<true>, dim3((unsigned)((count + kAmpEnvs - 1) / kAmpEnvs)), dim3(kAmpEnvs * kAmpLanes), 0, as_stream(s), a);
    } else {
        PULSE_REQUIRE(a.window_out == nullptr, "pulse_amp_obs: window_out goes with hist_steps > 1");
        hipLaunchKernelGGL(amp_obs_kernel<false>, dim3((unsigned)((count + kAmpEnvs - 1) / kAmpEnvs)), dim3(kAmpEnvs * kAmpLanes), 0, as_stream(s), a);
    }
    return check_launch("pulse_amp_obs");
}

int pulse_sizeof_amp_hist_args(void) { return (int)sizeof(pulse_amp_hist_args); }

int pulse_amp_hist_init(const pulse_amp_hist_args* args, pulse_stream_t s) {
    PULSE_REQUIRE(args != nullptr, "pulse_amp_hist_init: null args");
    const pulse_amp_hist_args& a = *args;
    PULSE_REQUIRE(a.num_envs >= 0 && a.hist_steps >= 1, "pulse_amp_hist_init: bad sizes");
    if (a.num_envs == 0 || a.hist_steps == 1) return PULSE_OK;
    const pulse_motion_tables& T = a.tab;
    PULSE_REQUIRE(T.frames && T.motion_lengths && T.motion_dt && T.motion_num_frames && T.length_starts, "pulse_amp_hist_init: null table pointer");
    PULSE_REQUIRE(T.num_bodies >= 1 && T.num_bodies <= kAmpLanes && T.frame_stride % 4 == 0 && T.off_grs % 4 == 0 && T.off_lrs % 4 == 0,
                  "pulse_amp_hist_init: bad motion tables");
    PULSE_REQUIRE(a.motion_ids && a.start_times && a.hist && a.key_body_ids, "pulse_amp_hist_init: null pointer");
    PULSE_REQUIRE(a.num_joints >= 1 && a.num_joints <= kAmpLanes && (a.joint_ids || a.num_joints <= T.num_bodies - 1) &&
                  a.num_key_bodies >= 0 && a.num_key_bodies <= kAmpLanes, "pulse_amp_hist_init: joints / key bodies must fit 32 lanes and the skeleton");
    const int w = pulse_amp_obs_width(a.num_joints, a.num_key_bodies, a.root_height_obs);
    PULSE_REQUIRE(w <= kAmpMaxW && a.step_stride >= w && a.env_stride >= (int64_t)a.hist_steps * a.step_stride, "pulse_amp_hist_init: bad pitches");
    const long long total = (long long)a.num_envs * (a.hist_steps - 1);
    hipLaunchKernelGGL(amp_hist_init_kernel, dim3((unsigned)((total + kAmpEnvs - 1) / kAmpEnvs)), dim3(kAmpEnvs * kAmpLanes), 0, as_stream(s), a);
    return check_launch("pulse_amp_hist_init");
}
}
