// Fused HumanoidIm post-physics step for gfx950: imitation reward -> reset -> next observation.
//
// Replaces, in one launch, the reference's chain of TorchScript functions
//   compute_imitation_reward            phc/env/tasks/humanoid_im.py:1543-1574 (+ power term :908-917)
//   compute_humanoid_im_reset           phc/env/tasks/humanoid_im.py:1600-1628 (+ recovery mask :1188-1190)
//   compute_humanoid_observations_smpl_max   phc/env/tasks/humanoid.py:1675-1731
//   compute_imitation_observations_v6 / _v7  phc/env/tasks/humanoid_im.py:1328-1413
// (each of which is O(30-70) elementwise launches in the reference's GPU pipeline).
//
// Mapping (MI355X-first, HBM/launch bound: ~8 KB algorithmic traffic per env-step):
//   * one 32-lane half-wave per environment, lane b <-> body b (24 of 32 lanes active);
//     ENVS_PER_BLOCK environments per workgroup, so N=4096 gives 1024 workgroups (>> 256 CUs);
//   * the env's 13-float AoS rigid-body records and both reference frames (t and t+1) are
//     staged into LDS with coalesced 16-byte loads, then read back at a 13-float lane stride
//     (odd stride -> conflict-free ds_read_b32);
//   * per-env reductions (4 reward errors, power, any-body-fell) are 32-lane butterfly shuffles;
//   * the 934-float observation row is assembled in LDS and written with full-line 16-byte
//     stores straight into the caller's row pitch (e.g. a 960-float GEMM-ready pitch, pad zeroed).
// Compiled with -ffp-contract=off so products/sums round exactly like the eager reference.
#include <cstdlib>
#include "common.h"
#include "rot_math.h"
#include "motion_math.h"

namespace pulse {

constexpr int kLanesPerEnv = 32;

struct ObsLayout {
    int self_w;      // width of the self observation (all history steps / force-sensor rows included)
    int self_step;   // width of one history step of the self observation
    int off_pos, off_rot, off_vel, off_ang;
    int task_w;      // width of the task observation
    int per_t;       // task obs floats per future sample (t-major variants 6 / 7 / 9)
};

// task observation widths, humanoid_im.py:452-496: v1 15 Jt T, v2 + 3 (Jt - 1), v3 9 Jt T, v6 24 Jt T, v7 9 Jt T, v8 30 Jt (T = 1),
// v9 (18 Jt + 6) T
__host__ __device__ inline int task_per_t(int v, int Jt) {
    return v == 7 ? 9 * Jt : v == 9 ? 18 * Jt + 6 : v == 6 ? 24 * Jt : v == 1 ? 15 * Jt : v == 2 ? 15 * Jt + 3 * (Jt - 1) : v == 3 ? 9 * Jt : 30 * Jt;
}

__host__ __device__ inline ObsLayout make_layout(int J, int root_height_obs, int obs_version, int Jt, int T, int self_v = 1, int hist = 1,
                                                 int fs_w = 0) {
    ObsLayout L;
    const int h0 = root_height_obs ? 1 : 0;
    L.off_pos = h0;
    L.off_rot = h0 + 3 * (J - 1);
    L.off_vel = L.off_rot + 6 * J;
    L.off_ang = L.off_vel + 3 * J;
    L.self_step = L.off_ang + 3 * J;
    L.self_w = self_v == 2 ? L.self_step * hist : L.self_step + fs_w;      // fs_w: every appended row (force sensors, shape, limb weights)
    L.per_t = task_per_t(obs_version, Jt);
    L.task_w = L.per_t * T;
    return L;
}

// offset of block X of tracked body j, future sample t inside the task observation.  Block ids: 0 diff pos (3), 1 diff rot (6),
// 2 diff vel (3), 3 diff ang vel (3), 4 ref pos (3), 5 ref rot (6), 6 ref vel (3), 7 ref ang vel (3); -1 = not in this version.
__device__ __forceinline__ int task_off(int v, int blk, int Jt, int T, int t, int j) {
    const int w = (blk == 1 || blk == 5) ? 6 : 3;
    if (v == 6) { const int base[8] = {0, 3, 9, 12, 15, 18, -1, -1}; return base[blk] < 0 ? -1 : t * 24 * Jt + base[blk] * Jt + w * j; }
    if (v == 7) { const int base[8] = {0, -1, 3, -1, 6, -1, -1, -1}; return base[blk] < 0 ? -1 : t * 9 * Jt + base[blk] * Jt + w * j; }
    if (v == 9) { const int base[8] = {0, 3, -1, -1, 9, 12, -1, -1}; return base[blk] < 0 ? -1 : t * (18 * Jt + 6) + base[blk] * Jt + (blk >= 4 ? 6 : 0) + w * j; }
    if (v == 8) { const int base[8] = {0, 3, 9, 12, 15, 18, 24, 27}; return (blk < 4 && t > 0) ? -1 : base[blk] * Jt + w * j; }
    // block-major over all samples: v1 / v2 (A B C D), v3 (A B)
    const int base[8] = {0, 3, 9, 12, -1, -1, -1, -1};
    if (base[blk] < 0 || (v == 3 && blk > 1)) return -1;
    return base[blk] * Jt * T + w * (t * Jt + j);
}

__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, kLanesPerEnv);
    return v;
}
__device__ __forceinline__ int group_or(int v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v |= __shfl_xor(v, o, kLanesPerEnv);
    return v;
}

// cooperative copy of n floats global -> LDS by the nl lanes of one env group (lane = 0 .. nl - 1)
__device__ __forceinline__ void stage(float* __restrict__ dst, const float* __restrict__ src, int n, int lane, bool vec_ok, int nl) {
    if (vec_ok) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        const int n4 = n >> 2;
        for (int i = lane; i < n4; i += nl) d4[i] = s4[i];
        for (int i = (n4 << 2) + lane; i < n; i += nl) dst[i] = src[i];
    } else {
        for (int i = lane; i < n; i += nl) dst[i] = src[i];
    }
}

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// R = half-waves per env.  R = 1: the 32 lanes of an env do everything in turn.  R = 2 (64 lanes per env): the env's two half-waves share the
// staging copies and split the arithmetic -- half 0: reference blend at t, self observation, reward, reset; half 1: reference blend(s) at
// t + 1, task observation -- so the dependent chain of a step (the launch is latency-bound: every env's waves are resident at once) is about
// half as long and the chip holds twice the waves.  Same operations on the same operands per output element: results are bit-identical.
template <int E, int R>
__global__ void __launch_bounds__(E * R * kLanesPerEnv) im_step_kernel(const pulse_im_step_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int J = a.num_bodies;
    const int J13 = J * 13;
    const int T = a.time_steps;
    const int nd = a.num_dof;
    const int J13p = (J13 + 3) & ~3;          // keep every LDS segment 16-byte aligned
    const int ndp = (nd + 3) & ~3;
    const int colsp = (a.what & (PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS)) ? ((a.obs_cols + 3) & ~3) : 0;   // as the host sizes the launch
    // LDS carve-up (floats); every base a multiple of 4 floats
    float* s_rb = smem;                        // [E][J13p]
    float* s_rn = s_rb + E * J13p;             // [E][J13p]   ref at t   : pos|rot|vel|ang
    float* s_rx = s_rn + E * J13p;             // [E][T][J13p] ref at t+1
    float* s_df = s_rx + E * T * J13p;         // [E][ndp]
    float* s_dv = s_df + E * ndp;              // [E][ndp]
    float* s_obs = s_dv + E * ndp;             // [E][colsp]
    float* s_rd = s_obs + E * colsp;           // [E][ndp]    reference dof positions (obs_version 2 only)

    const int tid = threadIdx.x;
    if (a.what & PULSE_IM_DEBUG_POISON_LDS) {      // debug aid: a read of LDS nobody wrote shows up as NaN in the outputs
        const int total = E * ((2 + T) * J13p + 3 * ndp + colsp);
        for (int i = tid; i < total; i += E * R * kLanesPerEnv) smem[i] = __int_as_float(0x7fc00000);
        __syncthreads();
    }
    constexpr int NL = R * kLanesPerEnv;           // lanes per env
    // the two half-waves of an env sit in DIFFERENT waves (a wave holds the same half of two envs): a wave never diverges between the halves'
    // code paths.  Threads [0, 32 E) are half 0 of envs 0 .. E-1, threads [32 E, 64 E) half 1.
    const int role = tid / (E * kLanesPerEnv);     // which half-wave of the env
    const int slot = (tid % (E * kLanesPerEnv)) / kLanesPerEnv;
    const int lane = tid % kLanesPerEnv;           // body / joint index inside the half-wave
    const int lane_all = role * kLanesPerEnv + lane;   // cooperative copies: every lane of the env
    const bool r_now = role == 0, r_next = role == R - 1;      // who blends / computes what (R = 1: one half-wave does both)
    const int count = a.env_ids ? a.num_ids : a.num_envs;
    const int idx = blockIdx.x * E + slot;
    bool valid = idx < count;
    int64_t e = 0;
    if (valid) {
        e = a.env_ids ? a.env_ids[idx] : (int64_t)idx;
        if (a.env_mask && a.env_mask[e] == 0) valid = false;
    }

    // episode clock: the progress value this launch works with (optionally advanced here) and the time-out flag
    long long prog = 0;
    bool pass_time = false;
    if (valid) {
        if (a.progress_rw) prog = a.progress_rw[e] + a.progress_inc;
        else if (a.progress) prog = a.progress[e];
        if (a.clock_motion_len || a.cycle_motion) {
            if (a.cycle_motion) {
                pass_time = prog >= (long long)a.max_episode_length - 1;
            } else {
                float t = (float)prog * a.clock_dt;
                if (a.clock_start_times) t = t + a.clock_start_times[e];
                if (a.clock_start_offsets) t = t + a.clock_start_offsets[e];
                pass_time = t >= a.clock_motion_len[e];
            }
        } else if (a.pass_time) {
            pass_time = a.pass_time[e] != 0;
        }
    }

    // HumanoidImGetup._compute_reset (humanoid_im_getup.py:203-210): a recovering env's clock does not advance -- reward and the reset
    // test use p, the value written back and the observation stage use p - 1
    const bool recovering = valid && (a.what & PULSE_IM_RESET) && a.recovery_counter && a.recovery_counter[e] > 0;
    const long long prog_obs = prog - (recovering ? 1 : 0);
    const bool do_self = a.what & PULSE_IM_SELF_OBS;
    const bool do_task = a.what & PULSE_IM_TASK_OBS;
    const bool do_rew = a.what & PULSE_IM_REWARD;
    const bool do_rst = a.what & PULSE_IM_RESET;
    const bool need_now = do_rew || do_rst;
    const int H = a.self_obs_version == 2 ? a.hist_steps : 1;
    // rows appended to the self observation: force-sensor readings (version 3), then shape, then limb-weight parameters
    const int fs_w = a.self_obs_version == 3 ? a.force_sensor_width : 0;
    const int app_w = fs_w + (a.smpl_params ? a.smpl_params_width : 0) + (a.limb_weights ? a.limb_weights_width : 0);
    const ObsLayout L = make_layout(J, a.root_height_obs, a.obs_version, a.num_track, T, a.self_obs_version, H, app_w);

    float* rb_e = s_rb + slot * J13p;
    float* rn_e = s_rn + slot * J13p;
    float* rx_e = s_rx + slot * T * J13p;
    float* df_e = s_df + slot * ndp;
    float* dv_e = s_dv + slot * ndp;
    float* obs_e = s_obs + slot * colsp;
    float* rd_e = s_rd + slot * ndp;

    // zero_out_far: the reward stage pays for the approach since the LAST observation (_point_goal); the task-observation stage of this very
    // launch overwrites it after the barrier, so every lane takes its copy now
    float prev_goal = 0.f;
    if (valid && a.zero_out_far && do_rew) prev_goal = a.point_goal[e];

    // ---------------- stage inputs (global -> LDS, coalesced) ----------------
    if (valid) {
        const float* g_rb = a.rb + e * a.rb_env_stride + (H - 1) * J13;      // the newest record of the history
        stage(rb_e, g_rb, J13, lane_all, aligned16(g_rb), NL);
        if (a.use_motion) {
            // reference motion straight from the packed library: lane j blends body j of the two frame records
            // (get_motion_state, motion_lib_base.py:434-517) into the same LDS images the array path stages
            const pulse_motion_tables& M = a.motion;
            const long long m = a.motion_ids[e];
            const float* off = a.motion_offset ? a.motion_offset + 3 * e : nullptr;
            if (need_now && r_now) {
                float t = (float)prog * a.clock_dt;                       // humanoid_im.py:859
                if (a.clock_start_times) t = t + a.clock_start_times[e];
                if (a.clock_start_offsets) t = t + a.clock_start_offsets[e];
                const FramePair fp = frame_pair(M, m, t);
                if (lane < J) {
                    const BodyState b = blend_body(M, fp.r0, fp.r1, fp.blend, lane, off);
                    float* o = rn_e + 3 * lane;           o[0] = b.p.x; o[1] = b.p.y; o[2] = b.p.z;
                    o = rn_e + J * 3 + 4 * lane;          o[0] = b.q.x; o[1] = b.q.y; o[2] = b.q.z; o[3] = b.q.w;
                    o = rn_e + J * 7 + 3 * lane;          o[0] = b.v.x; o[1] = b.v.y; o[2] = b.v.z;
                    o = rn_e + J * 10 + 3 * lane;         o[0] = b.w.x; o[1] = b.w.y; o[2] = b.w.z;
                }
            }
            if (do_task && r_next) {
                for (int k = 0; k < T; ++k) {
                    float t = (float)(prog_obs + 1) * a.clock_dt;         // humanoid_im.py:723-731 (next frame, so +1)
                    if (T > 1) t = t + (float)k * a.traj_dt;
                    if (a.clock_start_times) t = t + a.clock_start_times[e];
                    if (a.clock_start_offsets) t = t + a.clock_start_offsets[e];
                    const FramePair fp = frame_pair(M, m, t);
                    float* d = rx_e + k * J13p;
                    if (lane < J) {
                        const BodyState b = blend_body(M, fp.r0, fp.r1, fp.blend, lane, off);
                        float* o = d + 3 * lane;              o[0] = b.p.x; o[1] = b.p.y; o[2] = b.p.z;
                        o = d + J * 3 + 4 * lane;             o[0] = b.q.x; o[1] = b.q.y; o[2] = b.q.z; o[3] = b.q.w;
                        o = d + J * 7 + 3 * lane;             o[0] = b.v.x; o[1] = b.v.y; o[2] = b.v.z;
                        o = d + J * 10 + 3 * lane;            o[0] = b.w.x; o[1] = b.w.y; o[2] = b.w.z;
                        if (k == 0 && a.track_rb) {
                            float* rec = a.track_rb + e * a.track_rb_stride + 13 * lane;
                            rec[0] = b.p.x; rec[1] = b.p.y; rec[2] = b.p.z;
                            rec[3] = b.q.x; rec[4] = b.q.y; rec[5] = b.q.z; rec[6] = b.q.w;
                            rec[7] = b.v.x; rec[8] = b.v.y; rec[9] = b.v.z;
                            rec[10] = b.w.x; rec[11] = b.w.y; rec[12] = b.w.z;
                        }
                    }
                    if (k == 0 && (a.track_dof_pos || a.obs_version == 2) && lane < J - 1) {
                        V3 dp, dv;
                        blend_dof(M, fp.r0, fp.r1, fp.blend, lane, &dp, &dv);
                        if (a.track_dof_pos) {
                            float* o = a.track_dof_pos + e * (3 * (J - 1)) + 3 * lane;  o[0] = dp.x; o[1] = dp.y; o[2] = dp.z;
                            o = a.track_dof_vel + e * (3 * (J - 1)) + 3 * lane;         o[0] = dv.x; o[1] = dv.y; o[2] = dv.z;
                        }
                        if (a.obs_version == 2) { float* o = rd_e + 3 * lane; o[0] = dp.x; o[1] = dp.y; o[2] = dp.z; }
                    }
                }
            }
        } else {
            if (need_now) {
                // ref_now_* rows are J*3 / J*4 floats per env; 16-byte aligned whenever J % 4 == 0
                const float* p = a.ref_now_pos + e * (J * 3);
                const float* q = a.ref_now_rot + e * (J * 4);
                const float* v = a.ref_now_vel + e * (J * 3);
                const float* w = a.ref_now_ang + e * (J * 3);
                stage(rn_e, p, J * 3, lane_all, aligned16(p), NL);
                stage(rn_e + J * 3, q, J * 4, lane_all, aligned16(q) && ((J * 3) % 4 == 0), NL);
                stage(rn_e + J * 7, v, J * 3, lane_all, aligned16(v) && ((J * 7) % 4 == 0), NL);
                stage(rn_e + J * 10, w, J * 3, lane_all, aligned16(w) && ((J * 10) % 4 == 0), NL);
            }
            if (do_task) {
                for (int t = 0; t < T; ++t) {
                    const int64_t r = e * T + t;
                    float* d = rx_e + t * J13p;
                    const float* p = a.ref_next_pos + r * (J * 3);
                    const float* v = a.ref_next_vel + r * (J * 3);
                    stage(d, p, J * 3, lane_all, aligned16(p), NL);
                    stage(d + J * 7, v, J * 3, lane_all, aligned16(v) && ((J * 7) % 4 == 0), NL);
                    if (a.obs_version != 7) {
                        const float* q = a.ref_next_rot + r * (J * 4);
                        const float* w = a.ref_next_ang + r * (J * 3);
                        stage(d + J * 3, q, J * 4, lane_all, aligned16(q) && ((J * 3) % 4 == 0), NL);
                        stage(d + J * 10, w, J * 3, lane_all, aligned16(w) && ((J * 10) % 4 == 0), NL);
                    }
                }
                if (a.obs_version == 2) stage(rd_e, a.ref_next_dof_pos + e * nd, nd, lane_all, false, NL);
            }
        }
        if (do_rew && a.specs.power_reward) {
            const float* f = a.dof_force + e * nd;
            const float* v = a.dof_vel + e * nd;
            stage(df_e, f, nd, lane_all, aligned16(f), NL);
            stage(dv_e, v, nd, lane_all, aligned16(v), NL);
        }
        // zero the padding columns of the observation row
        if (do_self || do_task) {
            const int obs_w = L.self_w + L.task_w;
            for (int c = obs_w + lane_all; c < colsp; c += NL) obs_e[c] = 0.0f;
        }
    }
    __syncthreads();

    // ---------------- per-(env, body) math ----------------
    if (valid) {
        if (lane_all == 0) {                               // every lane of this env read the old value before the barrier
            if (a.progress_rw) a.progress_rw[e] = prog_obs;
            if (a.pass_time_out) a.pass_time_out[e] = pass_time ? 1 : 0;
        }
        const V3 root_p{rb_e[0], rb_e[1], rb_e[2]};
        Q4 root_q{rb_e[3], rb_e[4], rb_e[5], rb_e[6]};
        if (!a.upright_start) root_q = qmul(root_q, Q4{-0.5f, -0.5f, -0.5f, 0.5f});   // remove_base_rot, humanoid.py:1616-1620
        const Q4 hinv = heading_quat(root_q, true);   // calc_heading_quat_inv
        // calc_heading_quat is its exact conjugate: quat_from_angle_axis(+-h, z^) differ in the sign of sin(h / 2) only (sin is odd in fp32 too,
        // the normalisation is the same) -- one heading evaluation (atan2 + sin + cos + two normalisations) per lane instead of two [r6]
        const Q4 hfwd = qconj(hinv);
        const float h_s = 2.0f * (hinv.w * hinv.w) - 1.0f;                  // shared by every rotation into the heading frame (qrot_heading)

        if (do_self && r_now) {
            for (int hs = 0; hs < H; ++hs) {
                // history steps older than the newest are read straight from global memory (13 floats per lane)
                const float* src = hs == H - 1 ? rb_e : a.rb + e * a.rb_env_stride + hs * J13;
                float* ob = obs_e + hs * L.self_step;
                if (lane < J) {
                    const float* r = src + 13 * lane;
                    const V3 p{r[0], r[1], r[2]};
                    const Q4 q{r[3], r[4], r[5], r[6]};
                    const V3 v{r[7], r[8], r[9]};
                    const V3 w{r[10], r[11], r[12]};
                    if (lane >= 1) {
                        // every step is expressed relative to the NEWEST root (humanoid.py:1737-1753)
                        const V3 lp = qrot_heading(hinv, h_s, V3{p.x - root_p.x, p.y - root_p.y, p.z - root_p.z});
                        float* o = ob + L.off_pos + 3 * (lane - 1);
                        o[0] = lp.x; o[1] = lp.y; o[2] = lp.z;
                    } else if (a.root_height_obs) {
                        ob[0] = p.z;
                    }
                    float tn[6];
                    if (lane == 0 && !a.local_root_obs) q_to_tan_norm(root_q, tn);   // humanoid.py:1707-1709 (after remove_base_rot)
                    else q_to_tan_norm(qmul(hinv, q), tn);
                    float* o = ob + L.off_rot + 6 * lane;
#pragma unroll
                    for (int k = 0; k < 6; ++k) o[k] = tn[k];
                    const V3 lv = qrot_heading(hinv, h_s, v);
                    o = ob + L.off_vel + 3 * lane;
                    o[0] = lv.x; o[1] = lv.y; o[2] = lv.z;
                    const V3 lw = qrot_heading(hinv, h_s, w);
                    o = ob + L.off_ang + 3 * lane;
                    o[0] = lw.x; o[1] = lw.y; o[2] = lw.z;
                }
            }
            if (a.self_obs_version == 3)                            // force-sensor readings appended (humanoid.py:1838)
                for (int c = lane; c < a.force_sensor_width; c += kLanesPerEnv)
                    obs_e[L.self_step + c] = a.force_sensor[e * a.force_sensor_width + c];
            if (a.smpl_params)                                      // has_smpl_params, then has_limb_weight_params (humanoid.py:1724-1728)
                for (int c = lane; c < a.smpl_params_width; c += kLanesPerEnv)
                    obs_e[L.self_step + fs_w + c] = a.smpl_params[e * a.smpl_params_stride + c];
            if (a.limb_weights)
                for (int c = lane; c < a.limb_weights_width; c += kLanesPerEnv)
                    obs_e[L.self_step + fs_w + (a.smpl_params ? a.smpl_params_width : 0) + c] = a.limb_weights[e * a.limb_weights_stride + c];
        }

        if (do_task && r_next && lane < a.num_track) {
            const int Jt = a.num_track, ov = a.obs_version;
            const int tb = a.track_ids[lane];
            const float* r = rb_e + 13 * tb;
            const V3 p{r[0], r[1], r[2]};
            const Q4 q{r[3], r[4], r[5], r[6]};
            const V3 v{r[7], r[8], r[9]};
            const V3 w{r[10], r[11], r[12]};
            float* tob = obs_e + L.self_w;
            auto put3 = [&](int off, const V3& x) { if (off >= 0) { float* o = tob + off; o[0] = x.x; o[1] = x.y; o[2] = x.z; } };
            auto put6 = [&](int off, const float* x) { if (off >= 0) { float* o = tob + off; for (int k = 0; k < 6; ++k) o[k] = x[k]; } };
            // zero_out_far (humanoid_im.py:763-777, 814-826; T == 1): distance of the simulated root to the reference of tracked body 0
            bool zof_far = false, zof_dir = false;
            float zof_d = 0.f;
            if (a.zero_out_far) {
                const int t0 = a.track_ids[0];
                const float dx = root_p.x - rx_e[3 * t0], dy = root_p.y - rx_e[3 * t0 + 1], dz = root_p.z - rx_e[3 * t0 + 2];
                zof_d = sqrtf(dx * dx + dy * dy + dz * dz);
                zof_far = zof_d > a.close_distance;
                zof_dir = zof_d > a.far_distance;
                if (lane == 0) a.point_goal[e] = zof_d;
            }
            for (int t = 0; t < T; ++t) {
                const float* x = rx_e + t * J13p;
                V3 pr{x[3 * tb], x[3 * tb + 1], x[3 * tb + 2]};
                V3 vr{x[J * 7 + 3 * tb], x[J * 7 + 3 * tb + 1], x[J * 7 + 3 * tb + 2]};
                if (zof_far) {                       // a far env's reference is its own state ...
                    vr = v;
                    if (lane >= 1) pr = p;
                }
                const bool occl = a.occl_bits && ((a.occl_bits[e] >> lane) & 1u);       // occl_training (humanoid_im.py:778-784, 827-831), after the far masking
                if (occl) {
                    pr = p;
                    if (a.obs_version != 7) vr = v;
                }
                if (zof_dir && lane == 0 && !occl)   // ... and beyond far_distance the root target is only a direction
                    pr = V3{(pr.x - p.x) / zof_d * a.far_distance + p.x, (pr.y - p.y) / zof_d * a.far_distance + p.y,
                            (pr.z - p.z) / zof_d * a.far_distance + p.z};
                put3(task_off(ov, 0, Jt, T, t, lane), qrot_heading(hinv, h_s, V3{pr.x - p.x, pr.y - p.y, pr.z - p.z}));
                put3(task_off(ov, 2, Jt, T, t, lane), qrot_heading(hinv, h_s, V3{vr.x - v.x, vr.y - v.y, vr.z - v.z}));
                put3(task_off(ov, 4, Jt, T, t, lane), qrot_heading(hinv, h_s, V3{pr.x - root_p.x, pr.y - root_p.y, pr.z - root_p.z}));
                put3(task_off(ov, 6, Jt, T, t, lane), qrot_heading(hinv, h_s, vr));
                if (ov != 7) {
                    Q4 qr{x[J * 3 + 4 * tb], x[J * 3 + 4 * tb + 1], x[J * 3 + 4 * tb + 2], x[J * 3 + 4 * tb + 3]};
                    V3 wr{x[J * 10 + 3 * tb], x[J * 10 + 3 * tb + 1], x[J * 10 + 3 * tb + 2]};
                    if (zof_far) {
                        wr = w;
                        if (lane >= 1) qr = q;
                    }
                    if (occl) { qr = q; wr = w; }
                    float tn[6];
                    int off = task_off(ov, 1, Jt, T, t, lane);
                    if (off >= 0) { q_to_tan_norm(qmul(qmul(hinv, qmul(qr, qconj(q))), hfwd), tn); put6(off, tn); }   // change of basis
                    put3(task_off(ov, 3, Jt, T, t, lane), qrot_heading(hinv, h_s, V3{wr.x - w.x, wr.y - w.y, wr.z - w.z}));
                    off = task_off(ov, 5, Jt, T, t, lane);
                    if (off >= 0) { q_to_tan_norm(qmul(hinv, qr), tn); put6(off, tn); }
                    put3(task_off(ov, 7, Jt, T, t, lane), qrot_heading(hinv, h_s, wr));
                    if (ov == 9 && lane == 0) {          // root velocity differences (tracked body 0), humanoid_im.py:1510-1517
                        const int base = t * (18 * Jt + 6) + 9 * Jt;
                        put3(base, qrot_heading(hinv, h_s, V3{vr.x - v.x, vr.y - v.y, vr.z - v.z}));
                        put3(base + 3, qrot_heading(hinv, h_s, V3{wr.x - w.x, wr.y - w.y, wr.z - w.z}));
                    }
                }
            }
            if (ov == 2 && lane >= 1) {                  // dof differences of the tracked joints (humanoid_im.py:755-758, 1293-1294)
                const int d0 = 3 * (tb - 1);
                const float* cur = a.dof_pos + e * nd + d0;
                const float* ref = rd_e + d0;
                float* o = tob + 15 * Jt * T + 3 * (lane - 1);
                o[0] = ref[0] - cur[0]; o[1] = ref[1] - cur[1]; o[2] = ref[2] - cur[2];
            }
        }

        if (do_rew && r_now) {
            // bodies entering the reward: all J (full-body) or the tracked subset (humanoid_im.py:886-899)
            // (the zero_out_far branch always takes the full body, humanoid_im.py:876-878)
            const bool full = a.full_body_reward || a.zero_out_far;
            const int nb = full ? J : a.num_track;
            float e_pos = 0.f, e_rot = 0.f, e_vel = 0.f, e_ang = 0.f;
            if (lane < nb) {
                const int b = full ? lane : a.track_ids[lane];
                const float* r = rb_e + 13 * b;
                const float* x = rn_e;
                float dx = x[3 * b] - r[0], dy = x[3 * b + 1] - r[1], dz = x[3 * b + 2] - r[2];
                e_pos = (dx * dx + dy * dy + dz * dz) / 3.0f;
                const Q4 q{r[3], r[4], r[5], r[6]};
                const Q4 qr{x[J * 3 + 4 * b], x[J * 3 + 4 * b + 1], x[J * 3 + 4 * b + 2], x[J * 3 + 4 * b + 3]};
                const float ang = q_to_angle_axis(qmul(qr, qconj(q)), nullptr);
                e_rot = ang * ang;
                dx = x[J * 7 + 3 * b] - r[7]; dy = x[J * 7 + 3 * b + 1] - r[8]; dz = x[J * 7 + 3 * b + 2] - r[9];
                e_vel = (dx * dx + dy * dy + dz * dz) / 3.0f;
                dx = x[J * 10 + 3 * b] - r[10]; dy = x[J * 10 + 3 * b + 1] - r[11]; dz = x[J * 10 + 3 * b + 2] - r[12];
                e_ang = (dx * dx + dy * dy + dz * dz) / 3.0f;
            }
            e_pos = group_sum(e_pos); e_rot = group_sum(e_rot);
            e_vel = group_sum(e_vel); e_ang = group_sum(e_ang);
            float pw = 0.f;
            if (a.specs.power_reward) {
                for (int d = lane; d < nd; d += kLanesPerEnv) pw += fabsf(df_e[d] * dv_e[d]);
                pw = group_sum(pw);
            }
            if (lane == 0) {
                const float fn = (float)nb;
                const float r_pos = expf(-a.specs.k_pos * (e_pos / fn));
                const float r_rot = expf(-a.specs.k_rot * (e_rot / fn));
                const float r_vel = expf(-a.specs.k_vel * (e_vel / fn));
                const float r_ang = expf(-a.specs.k_ang_vel * (e_ang / fn));
                float rew = a.specs.w_pos * r_pos + a.specs.w_rot * r_rot + a.specs.w_vel * r_vel + a.specs.w_ang_vel * r_ang;
                const int rw = a.specs.power_reward ? 5 : 4;
                float* raw = a.rew_raw + e * rw;
                if (a.zero_out_far) {
                    // _compute_reward's zero_out_far branch (humanoid_im.py:870-887): compute_point_goal_reward (:1577-1582) for everyone,
                    // half the imitation reward on top within transition_distance = 0.25 m of the reference root
                    const float dx = root_p.x - rn_e[0], dy = root_p.y - rn_e[1], dz = root_p.z - rn_e[2];
                    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
                    const float g = fminf(prev_goal - dist, (float)(1.0 / 3.0)) * 9.0f;
                    if (dist > 0.25f) {
                        rew = g;
                        raw[0] = g; raw[1] = 0.f; raw[2] = 0.f; raw[3] = 0.f;
                    } else {
                        rew = g + rew * 0.5f;
                        raw[0] = g + r_pos * 0.5f; raw[1] = r_rot * 0.5f; raw[2] = r_vel * 0.5f; raw[3] = r_ang * 0.5f;
                    }
                } else {
                    raw[0] = r_pos; raw[1] = r_rot; raw[2] = r_vel; raw[3] = r_ang;
                }
                if (a.specs.power_reward) {
                    float p = -a.specs.power_coef * pw;
                    if (prog <= 3) p = 0.0f;     // first frames are not charged (humanoid_im.py:914)
                    rew += p;
                    raw[4] = p;
                }
                a.rew[e] = rew;
            }
        }

        if (do_rst && r_now) {
            float dist = 0.f;
            int fell = 0;
            if (lane < a.num_reset) {
                const int b = a.reset_ids[lane];
                const float* r = rb_e + 13 * b;
                const float dx = r[0] - rn_e[3 * b], dy = r[1] - rn_e[3 * b + 1], dz = r[2] - rn_e[3 * b + 2];
                dist = sqrtf(dx * dx + dy * dy + dz * dz);
                // occl_training: an occluded reset body's reference is its own position (_compute_reset, humanoid_im.py:1178-1183; the mask is indexed by body id)
                if (a.occl_bits && a.occl_reset && ((a.occl_bits[e] >> b) & 1u)) dist = 0.f;
                fell = dist > a.term_dist[b];
            }
            int fallen;
            if (a.reset_use_mean) {
                const float m = group_sum(dist) / (float)a.num_reset;
                fallen = m > a.term_dist[a.reset_ids[0]];
            } else {
                fallen = group_or(fell);
            }
            if (lane == 0) {
                const bool pt = pass_time;
                int64_t term = (a.enable_early_termination && fallen && prog > 1) ? 1 : 0;
                int64_t rst = pt ? 1 : term;
                if (a.cycle_counter && !pt && a.cycle_counter[e] > 0) { rst = 0; term = 0; }
                if (recovering) { rst = 0; term = 0; }
                a.reset[e] = rst;
                a.terminate[e] = term;
            }
        }
    }
    if (!(do_self || do_task)) return;
    __syncthreads();

    // ---------------- write the observation row (LDS -> global, 16-byte stores) ----------------
    if (valid) {
        const int c0 = do_self ? 0 : L.self_w;
        const int c1 = do_task ? a.obs_cols : L.self_w;
        float* g = a.obs + e * a.obs_stride;
        float* g2 = a.obs_copy ? a.obs_copy + e * a.obs_copy_stride : nullptr;      // the rollout's second record of the row (v29)
        if (c0 == 0 && (c1 & 3) == 0 && aligned16(g) && (!g2 || aligned16(g2))) {
            const float4* s4 = reinterpret_cast<const float4*>(obs_e);
            float4* g4 = reinterpret_cast<float4*>(g);
            float4* h4 = reinterpret_cast<float4*>(g2);
            for (int i = lane_all; i < (c1 >> 2); i += NL) {
                const float4 v = s4[i];
                g4[i] = v;
                if (g2) h4[i] = v;
            }
        } else {
            for (int c = c0 + lane_all; c < c1; c += NL) {
                g[c] = obs_e[c];
                if (g2) g2[c] = obs_e[c];
            }
        }
    }
}

}  // namespace pulse

using namespace pulse;

// PULSE_IM_TWO_ROLES=0: one half-wave per env always (A/B switch, read once)
static const bool g_im_two_roles = [] { const char* v = getenv("PULSE_IM_TWO_ROLES"); return !(v && v[0] == '0'); }();

extern "C" {

int pulse_sizeof_im_step_args(void) { return (int)sizeof(pulse_im_step_args); }
int pulse_self_obs_width(int num_bodies, int root_height_obs) {
    return make_layout(num_bodies, root_height_obs, 6, 0, 1).self_w;
}
int pulse_self_obs_width_ex(int num_bodies, int root_height_obs, int self_obs_version, int hist_steps, int force_sensor_width) {
    return make_layout(num_bodies, root_height_obs, 6, 0, 1, self_obs_version, hist_steps, self_obs_version == 2 ? 0 : force_sensor_width).self_w;
}
int pulse_task_obs_width(int obs_version, int num_track, int time_steps) {
    return make_layout(1, 0, obs_version, num_track, time_steps).task_w;
}

int pulse_im_step(const pulse_im_step_args* args, pulse_stream_t s) {
    PULSE_REQUIRE(args != nullptr, "pulse_im_step: null args");
    const pulse_im_step_args& a = *args;
    PULSE_REQUIRE(a.num_envs >= 0, "pulse_im_step: negative num_envs");
    const int count = a.env_ids ? a.num_ids : a.num_envs;
    if (count == 0 || (a.what & 15u) == 0) return PULSE_OK;
    PULSE_REQUIRE(a.rb != nullptr, "pulse_im_step: null rb");
    PULSE_REQUIRE(a.num_bodies >= 1 && a.num_bodies <= kLanesPerEnv, "pulse_im_step: num_bodies %d not in [1,32]", a.num_bodies);
    PULSE_REQUIRE(a.rb_env_stride >= (int64_t)a.num_bodies * 13, "pulse_im_step: rb_env_stride too small");
    PULSE_REQUIRE(a.time_steps >= 1, "pulse_im_step: time_steps < 1");
    const bool do_obs = a.what & (PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS);
    PULSE_REQUIRE(a.self_obs_version >= 1 && a.self_obs_version <= 3, "pulse_im_step: self_obs_version %d unsupported (1|2|3)", a.self_obs_version);
    const int hist = a.self_obs_version == 2 ? a.hist_steps : 1;
    PULSE_REQUIRE(hist >= 1, "pulse_im_step: hist_steps < 1");
    PULSE_REQUIRE(a.rb_env_stride >= (int64_t)hist * a.num_bodies * 13, "pulse_im_step: rb_env_stride does not cover the history");
    PULSE_REQUIRE(a.self_obs_version != 3 || (a.force_sensor != nullptr && a.force_sensor_width >= 0), "pulse_im_step: self_obs_version 3 needs force_sensor");
    PULSE_REQUIRE((a.smpl_params == nullptr && a.limb_weights == nullptr) || a.self_obs_version != 2,
                  "pulse_im_step: shape / limb-weight rows are not defined for self_obs_version 2 (the reference raises, humanoid.py:1780-1784)");
    PULSE_REQUIRE((!a.smpl_params || (a.smpl_params_width >= 0 && a.smpl_params_stride >= a.smpl_params_width)) &&
                  (!a.limb_weights || (a.limb_weights_width >= 0 && a.limb_weights_stride >= a.limb_weights_width)), "pulse_im_step: bad shape / limb-weight rows");
    PULSE_REQUIRE(a.recovery_counter == nullptr || a.progress_rw != nullptr, "pulse_im_step: recovery_counter needs the in-kernel clock (progress_rw)");
    const int app_w = (a.self_obs_version == 3 ? a.force_sensor_width : 0) + (a.smpl_params ? a.smpl_params_width : 0) + (a.limb_weights ? a.limb_weights_width : 0);
    const ObsLayout L = make_layout(a.num_bodies, a.root_height_obs, a.obs_version, a.num_track, a.time_steps, a.self_obs_version, hist, app_w);
    if (do_obs) {
        PULSE_REQUIRE(a.obs != nullptr, "pulse_im_step: null obs");
        PULSE_REQUIRE(a.obs_cols >= L.self_w + ((a.what & PULSE_IM_TASK_OBS) ? L.task_w : 0),
                      "pulse_im_step: obs_cols %d < observation width %d", a.obs_cols, L.self_w + L.task_w);
        PULSE_REQUIRE(a.obs_stride >= a.obs_cols, "pulse_im_step: obs_stride < obs_cols");
        PULSE_REQUIRE(a.obs_copy == nullptr || a.obs_copy_stride >= a.obs_cols, "pulse_im_step: obs_copy_stride < obs_cols");
    }
    if (a.what & PULSE_IM_TASK_OBS) {
        const int ov = a.obs_version;
        PULSE_REQUIRE(ov == 1 || ov == 2 || ov == 3 || ov == 6 || ov == 7 || ov == 8 || ov == 9, "pulse_im_step: obs_version %d unsupported (1|2|3|6|7|8|9)", ov);
        PULSE_REQUIRE((ov != 8 && ov != 2) || a.time_steps == 1, "pulse_im_step: obs_version %d takes one reference sample (the reference indexes a column for T > 1)", ov);
        if (ov == 2) {
            PULSE_REQUIRE(a.dof_pos && a.num_dof == 3 * (a.num_bodies - 1) && (a.use_motion || a.ref_next_dof_pos), "pulse_im_step: obs_version 2 needs dof_pos / ref_next_dof_pos");
        }
        PULSE_REQUIRE(a.track_ids != nullptr && a.num_track >= 1 && a.num_track <= a.num_bodies, "pulse_im_step: bad track ids");
        if (!a.use_motion) {
            PULSE_REQUIRE(a.ref_next_pos && a.ref_next_vel, "pulse_im_step: null ref_next");
            PULSE_REQUIRE(a.obs_version == 7 || (a.ref_next_rot && a.ref_next_ang), "pulse_im_step: null ref_next rot/ang");
        }
    }
    if (a.what & (PULSE_IM_REWARD | PULSE_IM_RESET)) {
        PULSE_REQUIRE(a.use_motion || (a.ref_now_pos && a.ref_now_rot && a.ref_now_vel && a.ref_now_ang), "pulse_im_step: null ref_now");
        PULSE_REQUIRE(a.progress != nullptr || a.progress_rw != nullptr, "pulse_im_step: null progress");
    }
    if (a.what & PULSE_IM_REWARD) {
        PULSE_REQUIRE(a.rew && a.rew_raw, "pulse_im_step: null reward outputs");
        PULSE_REQUIRE(a.full_body_reward || (a.track_ids && a.num_track >= 1), "pulse_im_step: subset reward needs track ids");
        if (a.specs.power_reward)
            PULSE_REQUIRE(a.dof_force && a.dof_vel && a.num_dof >= 1, "pulse_im_step: power reward needs dof force/vel");
    }
    if (a.what & PULSE_IM_RESET) {
        PULSE_REQUIRE(a.reset && a.terminate && a.term_dist, "pulse_im_step: null reset inputs/outputs");
        PULSE_REQUIRE(a.pass_time || a.clock_motion_len || a.cycle_motion, "pulse_im_step: reset needs pass_time or the in-kernel clock");
        PULSE_REQUIRE(a.reset_ids && a.num_reset >= 1 && a.num_reset <= kLanesPerEnv, "pulse_im_step: bad reset ids");
    }
    if (a.zero_out_far) {
        PULSE_REQUIRE(a.time_steps == 1, "pulse_im_step: zero_out_far takes one reference sample (the reference broadcasts (N, 3) against (N T, 3))");
        PULSE_REQUIRE(a.point_goal != nullptr, "pulse_im_step: zero_out_far needs point_goal");
        PULSE_REQUIRE(a.far_distance > 0.f && a.close_distance >= 0.f, "pulse_im_step: zero_out_far needs close_distance >= 0 and far_distance > 0");
        const int ov = a.obs_version;
        PULSE_REQUIRE(!(a.what & PULSE_IM_TASK_OBS) || ov == 6 || ov == 7 || ov == 8 || ov == 9,
                      "pulse_im_step: zero_out_far is defined for obs_version 6 | 7 | 8 | 9 (humanoid_im.py:761,812), not %d", ov);
    }
    if (a.occl_bits) {
        const int ov = a.obs_version;
        PULSE_REQUIRE(!(a.what & PULSE_IM_TASK_OBS) || ov == 6 || ov == 7 || ov == 8 || ov == 9,
                      "pulse_im_step: occl_bits is defined for obs_version 6 | 7 | 8 | 9 (humanoid_im.py:778,827), not %d", ov);
        PULSE_REQUIRE(a.time_steps == 1, "pulse_im_step: occl_bits takes one reference sample (an (N, Jt) mask indexes an (N T, Jt, .) reference in the reference)");
        PULSE_REQUIRE(!a.occl_reset || a.num_track == a.num_bodies, "pulse_im_step: occl_reset indexes the mask by body id: every body must be tracked, in order");
    }
    if (a.use_motion) {
        const pulse_motion_tables& M = a.motion;
        PULSE_REQUIRE(M.frames && M.motion_lengths && M.motion_dt && M.motion_num_frames && M.length_starts && a.motion_ids,
                      "pulse_im_step: in-kernel reference needs the motion tables and motion_ids");
        PULSE_REQUIRE(M.num_bodies == a.num_bodies, "pulse_im_step: motion library has %d bodies, env %d", M.num_bodies, a.num_bodies);
        PULSE_REQUIRE(M.frame_stride % 4 == 0 && M.off_grs % 4 == 0 && M.off_lrs % 4 == 0, "pulse_im_step: motion record fields must be 16-B aligned");
        PULSE_REQUIRE(a.progress != nullptr || a.progress_rw != nullptr, "pulse_im_step: in-kernel reference needs the episode clock (progress)");
        PULSE_REQUIRE(a.clock_dt > 0.f, "pulse_im_step: in-kernel reference needs clock_dt");
        PULSE_REQUIRE((a.track_dof_pos == nullptr) == (a.track_dof_vel == nullptr), "pulse_im_step: track_dof_pos / track_dof_vel go together");
        PULSE_REQUIRE(a.track_rb == nullptr || a.track_rb_stride >= 13 * a.num_bodies, "pulse_im_step: track_rb_stride too small");
    }
    constexpr int E = 4;
    const int J13p = (a.num_bodies * 13 + 3) & ~3;
    const int ndp = (a.num_dof + 3) & ~3;
    const int colsp = do_obs ? ((a.obs_cols + 3) & ~3) : 0;
    const size_t lds = sizeof(float) * (size_t)E * ((2 + a.time_steps) * J13p + 3 * ndp + colsp);
    PULSE_REQUIRE(lds <= 160 * 1024, "pulse_im_step: LDS request %zu > 160 KiB", lds);
    const unsigned grid = (unsigned)((count + E - 1) / E);
    // a whole wave per env when both halves have work (task observation beside self observation / reward / reset): see the kernel's note
    const bool two = (a.what & PULSE_IM_TASK_OBS) && (a.what & (PULSE_IM_SELF_OBS | PULSE_IM_REWARD | PULSE_IM_RESET)) && g_im_two_roles;
    const void* fn = two ? reinterpret_cast<const void*>(im_step_kernel<E, 2>) : reinterpret_cast<const void*>(im_step_kernel<E, 1>);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(PULSE_ERR_LAUNCH, "pulse_im_step: cannot raise LDS limit: %s", hipGetErrorString(e));
    }
    if (two) hipLaunchKernelGGL((im_step_kernel<E, 2>), dim3(grid), dim3(E * 2 * kLanesPerEnv), lds, as_stream(s), a);
    else hipLaunchKernelGGL((im_step_kernel<E, 1>), dim3(grid), dim3(E * kLanesPerEnv), lds, as_stream(s), a);
    return check_launch("pulse_im_step");
}
}
