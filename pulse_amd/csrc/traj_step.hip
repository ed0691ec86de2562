// Trajectory-following / terrain-traversal task (HumanoidTraj, HumanoidPedestrianTerrain) for gfx950: the last README command of the
// reference without a counterpart after round 2 (learning=pulse_z_terrain, network amp_sept).
//
// Replaces
//   TrajGenerator.reset / calc_pos                      phc/utils/traj_generator.py:60-123, 156-171
//   HumanoidTraj._fetch_traj_samples                    phc/env/tasks/humanoid_traj.py:196-211
//   compute_location_observations / _reward(_fuzzy)     phc/env/tasks/humanoid_pedestrian_terrain.py:1587-1646
//   get_heights / get_center_heights + Terrain.world_points_to_map / sample_height_points   :690-772, 1191-1270
//   _compute_task_obs (height clip / scale)             :384-440
//   compute_humanoid_reset (terrain and traj variants)  :1476-1531, humanoid_traj.py:256-300
//
// A different kernel style from the fused imitation step: the observation of ONE env is a 2-D table lookup of ~1000 height samples
// (32 x 32 sensor grid rotated by the head's heading, two int16 gathers each) plus ten trajectory samples, so the mapping is one
// 256-thread workgroup per env, four grid points per thread, coalesced stores straight into the env's GEMM-ready observation row.
// The height field (a few MB of int16) lives in L2; per env the kernel reads ~4 KB of it and writes 4.2 KB of observation.
// Reward and reset are a handful of scalars per env and ride in the same launch (thread 0 / one wave reduction).
// -ffp-contract=off: the expressions follow the reference's operation order.
#include "common.h"
#include "rot_math.h"

namespace pulse {

// isaacgym.torch_utils.quat_apply (3P): b + w t + xyz x t, t = 2 xyz x b
__device__ __forceinline__ V3 q_apply(const Q4 a, const V3 b) {
    const V3 t{(a.y * b.z - a.z * b.y) * 2.0f, (a.z * b.x - a.x * b.z) * 2.0f, (a.x * b.y - a.y * b.x) * 2.0f};
    return V3{b.x + a.w * t.x + (a.y * t.z - a.z * t.y), b.y + a.w * t.y + (a.z * t.x - a.x * t.z), b.z + a.w * t.z + (a.x * t.y - a.y * t.x)};
}

// TrajGenerator.calc_pos for one (trajectory, time)
__device__ __forceinline__ V3 traj_pos(const float* verts, int num_verts, float traj_dur, float time) {
    const int segs = num_verts - 1;
    float phase = time / traj_dur;
    phase = fminf(fmaxf(phase, 0.0f), 1.0f);
    const float seg = phase * (float)segs;
    const float f0 = floorf(seg), f1 = ceilf(seg);
    const float lerp = seg - f0;
    const float* p0 = verts + 3 * (int)f0;
    const float* p1 = verts + 3 * (int)f1;
    const float a = 1.0f - lerp;
    return V3{a * p0[0] + lerp * p1[0], a * p0[1] + lerp * p1[1], a * p0[2] + lerp * p1[2]};
}

// world_points_to_map: (points / horizontal_scale).long() truncates toward zero, then clip to [0, size - 2]; the two cells get_heights compares
__device__ __forceinline__ void height_cells(const pulse_traj_step_args& a, float wx, float wy, long long& c1, long long& c2) {
    long long px = (long long)(wx / a.horizontal_scale), py = (long long)(wy / a.horizontal_scale);
    px = px < 0 ? 0 : (px > a.map_rows - 2 ? a.map_rows - 2 : px);
    py = py < 0 ? 0 : (py > a.map_cols - 2 ? a.map_cols - 2 : py);
    c1 = px * a.map_cols + py;
    c2 = (px + 1) * a.map_cols + py + 1;
}
__device__ __forceinline__ float sample_height(const pulse_traj_step_args& a, float wx, float wy) {
    long long c1, c2;
    height_cells(a, wx, wy, c1, c2);
    const short h1 = a.heightsamples[c1], h2 = a.heightsamples[c2];
    return (float)(h1 < h2 ? h1 : h2) * a.vertical_scale;
}

__global__ void __launch_bounds__(256) traj_step_kernel(const pulse_traj_step_args a) {
    __shared__ float s_center;
    __shared__ float s_cpt[256];
    __shared__ float s_red[4][4];
    const int count = a.env_ids ? a.num_ids : a.num_envs;
    const int idx = blockIdx.x;
    if (idx >= count) return;
    const int64_t e = a.env_ids ? a.env_ids[idx] : (int64_t)idx;
    if (a.env_mask && a.env_mask[e] == 0) return;
    const int tid = threadIdx.x;
    const float* rb = a.rb + e * a.rb_env_stride;
    const V3 root_p{rb[0], rb[1], rb[2]};
    Q4 root_q{rb[3], rb[4], rb[5], rb[6]};
    if (!a.upright_start) root_q = qmul(root_q, Q4{-0.5f, -0.5f, -0.5f, 0.5f});          // remove_base_rot
    const float* verts = a.verts + e * (int64_t)a.num_verts * 3;
    const float time = (float)a.progress[e] * a.dt;                                       // progress_buf * self.dt

    if (a.what & PULSE_TASK_OBS) {
        float* o = a.obs + e * a.obs_stride + a.obs_offset;
        if (tid < a.num_samples) {                                                        // compute_location_observations
            const Q4 hinv = heading_quat(root_q, true);
            const float t = time + (float)tid * a.sample_timestep;
            const V3 p = traj_pos(verts, a.num_verts, a.traj_dur, t);
            const V3 l = qrot(hinv, V3{p.x - root_p.x, p.y - root_p.y, p.z - root_p.z});
            o[2 * tid] = l.x; o[2 * tid + 1] = l.y;
        }
        if (a.num_height_points > 0) {
            float* ho = o + 2 * a.num_samples;
            if (a.heightsamples == nullptr) {                                             // terrainType 'plane': zero heights
                for (int p = tid; p < a.num_height_points; p += 256) {
                    const float ref = a.use_center_height ? 0.0f : root_p.z;
                    ho[p] = fminf(fmaxf(ref - 0.0f, -3.0f), 3.0f) * a.height_meas_scale;
                }
            } else {
                if (a.use_center_height) {
                    // get_center_heights: 3 x 3 grid under the root, rotated by the root's yaw-only quaternion (quat_apply_yaw).  One thread per
                    // point (their dependent gathers overlap), summed by thread 0 in the reference's order
                    if (tid < a.num_center_points) {
                        const float n = fmaxf(sqrtf(root_q.z * root_q.z + root_q.w * root_q.w), 1e-9f);
                        const Q4 qy{0.0f / n, 0.0f / n, root_q.z / n, root_q.w / n};
                        const V3 w = q_apply(qy, V3{a.center_points[2 * tid], a.center_points[2 * tid + 1], 0.0f});
                        s_cpt[tid] = sample_height(a, w.x + root_p.x, w.y + root_p.y);
                    }
                    __syncthreads();
                    if (tid == 0) {
                        float sum = 0.0f;
                        for (int c = 0; c < a.num_center_points; ++c) sum += s_cpt[c];
                        s_center = sum / (float)a.num_center_points;
                    }
                }
                __syncthreads();
                // get_heights: the sensor grid rotated by the HEADING of the sensor body (terrain_obs_root: head) and moved to it
                const float* sb = rb + 13 * a.sensor_body;
                Q4 sq{sb[3], sb[4], sb[5], sb[6]};
                if (!a.upright_start) sq = qmul(sq, Q4{-0.5f, -0.5f, -0.5f, 0.5f});
                const Q4 hq = heading_quat(sq, false);
                const float ref = a.use_center_height ? s_center : root_p.z;
                // four points per thread and pass, their cell indices first and their eight gathers together: the gathers' latencies overlap
                for (int p0 = tid; p0 < a.num_height_points; p0 += 1024) {
                    long long i1[4], i2[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int p = min(p0 + 256 * i, a.num_height_points - 1);
                        const V3 w = q_apply(hq, V3{a.height_points[2 * p], a.height_points[2 * p + 1], 0.0f});
                        height_cells(a, w.x + sb[0], w.y + sb[1], i1[i], i2[i]);
                    }
                    short h1[4], h2[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { h1[i] = a.heightsamples[i1[i]]; h2[i] = a.heightsamples[i2[i]]; }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int p = p0 + 256 * i;
                        if (p < a.num_height_points) {
                            const float h = (float)(h1[i] < h2[i] ? h1[i] : h2[i]) * a.vertical_scale;
                            ho[p] = fminf(fmaxf(ref - h, -3.0f), 3.0f) * a.height_meas_scale;
                        }
                    }
                }
            }
        }
    }

    const bool need_tar = a.what & (PULSE_TASK_REWARD | PULSE_TASK_RESET);
    V3 tar{0.f, 0.f, 0.f};
    if (need_tar) tar = traj_pos(verts, a.num_verts, a.traj_dur, time);

    if (a.what & PULSE_TASK_REWARD) {
        // power = sum |tau qdot| over the dofs (all 256 threads), location reward by thread 0
        float pw = 0.0f;
        if (a.dof_force) {
            const float* f = a.dof_force + e * a.num_dof;
            const float* v = a.dof_vel + e * a.num_dof;
            for (int d = tid; d < a.num_dof; d += 256) pw += fabsf(f[d] * v[d]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pw += __shfl_xor(pw, o, 64);
        if ((tid & 63) == 0) s_red[0][tid >> 6] = pw;
        __syncthreads();
        if (tid == 0) {
            const float power = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
            const float dx = tar.x - root_p.x, dy = tar.y - root_p.y;
            float err = dx * dx + dy * dy;
            if (a.fuzzy_target && err < 0.0025f) err = 0.0f;
            const float loc = expf(-2.0f * err);
            const float pr = -a.power_coef * power;
            a.rew[e] = a.power_reward ? loc + pr : loc;
            if (a.rew_raw) { a.rew_raw[2 * e] = loc; a.rew_raw[2 * e + 1] = pr; }
        }
    }

    if ((a.what & PULSE_TASK_RESET) && tid == 0) {
        const long long prog = a.progress[e];
        int64_t term = 0;
        if (a.enable_early_termination) {
            const float* cf = a.contact_forces + e * (3 * a.num_bodies);
            bool fallen;
            if (a.terrain_reset) {
                // summed non-foot contact force above 50 N (humanoid_pedestrian_terrain.py:1490-1497)
                float sx = 0.f, sy = 0.f, sz = 0.f;
                for (int b = 0; b < a.num_bodies; ++b) {
                    bool foot = false;
                    for (int k = 0; k < a.num_contact_ids; ++k) foot |= a.contact_body_ids[k] == b;
                    if (foot) continue;
                    sx += cf[3 * b]; sy += cf[3 * b + 1]; sz += cf[3 * b + 2];
                }
                fallen = sqrtf(fabsf(sx) * fabsf(sx) + fabsf(sy) * fabsf(sy) + fabsf(sz) * fabsf(sz)) > 50.0f;
            } else {
                bool fall_contact = false, fall_height = false;
                for (int b = 0; b < a.num_bodies; ++b) {
                    bool foot = false;
                    for (int k = 0; k < a.num_contact_ids; ++k) foot |= a.contact_body_ids[k] == b;
                    if (foot) continue;
                    fall_contact |= fabsf(cf[3 * b]) > 0.1f || fabsf(cf[3 * b + 1]) > 0.1f || fabsf(cf[3 * b + 2]) > 0.1f;
                    fall_height |= rb[13 * b + 2] < a.termination_heights[b];
                }
                fallen = fall_contact && fall_height;
            }
            fallen = fallen && prog > 1;
            const float dx = tar.x - rb[0], dy = tar.y - rb[1];                           // (the un-rotated root position)
            const bool tar_fail = (dx * dx + dy * dy) > a.fail_dist * a.fail_dist;
            bool failed = fallen || tar_fail;
            if (a.disable_collision) failed = false;
            term = failed ? 1 : 0;
        }
        a.reset[e] = ((float)prog >= a.max_episode_length - 1.0f) ? 1 : term;
        a.terminate[e] = term;
    }
}

// TrajGenerator.reset for the masked envs: one thread per env walks the num_verts - 1 segments (speed scan with clipping, heading and
// vertex prefix sums are sequential by definition; 100 steps of a few flops).  The uniform draws come from the caller in the
// reference's order: u_dtheta, u_sharp, sharp_mask (Bernoulli), u_heading, u_dspeed, u_speed0.
__global__ void __launch_bounds__(64) traj_generate_kernel(const pulse_traj_gen_args a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.num_envs) return;
    if (a.env_mask && a.env_mask[e] == 0) return;
    const int segs = a.num_verts - 1;
    const float* rb = a.rb + (int64_t)e * a.rb_env_stride;
    float* v = a.verts + (int64_t)e * a.num_verts * 3;
    const float* ud = a.u_dtheta + (int64_t)e * segs;
    const float* us = a.u_sharp + (int64_t)e * segs;
    const uint8_t* sm = a.sharp_mask + (int64_t)e * segs;
    const float* uv = a.u_dspeed + (int64_t)e * segs;
    const float pi = 3.14159265358979323846f;
    float theta = 0.f, speed = 0.f, x = 0.f, y = 0.f;
    v[0] = rb[0]; v[1] = rb[1]; v[2] = 0.f;
    for (int i = 0; i < segs; ++i) {
        float dth = (2.0f * ud[i] - 1.0f) * a.dtheta_scale;
        if (sm[i]) dth = pi * (2.0f * us[i] - 1.0f);
        if (i == 0) dth = pi * (2.0f * a.u_heading[e] - 1.0f);
        const float dsp = (2.0f * uv[i] - 1.0f) * a.dspeed_scale;
        if (i == 0) speed = (a.speed_max - a.speed_min) * a.u_speed0[e] + a.speed_min;
        else speed = fminf(fmaxf(speed + dsp, a.speed_min), a.speed_max);
        theta = i == 0 ? dth : theta + dth;                                               // torch.cumsum
        const float len = speed * a.seg_dt;
        float dx = cosf(theta) * len, dy = -sinf(theta) * len;
        if (i == 0) { dx = dx + rb[0]; dy = dy + rb[1]; x = dx; y = dy; }
        else { x = x + dx; y = y + dy; }
        v[3 * (i + 1)] = x; v[3 * (i + 1) + 1] = y; v[3 * (i + 1) + 2] = 0.f;
    }
}

}  // namespace pulse

using namespace pulse;

extern "C" {

int pulse_sizeof_traj_step_args(void) { return (int)sizeof(pulse_traj_step_args); }

int pulse_traj_step(const pulse_traj_step_args* args, pulse_stream_t s) {
    PULSE_REQUIRE(args != nullptr, "pulse_traj_step: null args");
    const pulse_traj_step_args& a = *args;
    PULSE_REQUIRE(a.num_envs >= 0, "pulse_traj_step: negative num_envs");
    const int count = a.env_ids ? a.num_ids : a.num_envs;
    if (count == 0 || a.what == 0) return PULSE_OK;
    PULSE_REQUIRE(a.rb && a.num_bodies >= 1 && a.rb_env_stride >= 13LL * a.num_bodies, "pulse_traj_step: bad rigid-body state");
    PULSE_REQUIRE(a.verts && a.num_verts >= 2 && a.traj_dur > 0.f && a.progress && a.dt > 0.f, "pulse_traj_step: trajectory table / clock missing");
    if (a.what & PULSE_TASK_OBS) {
        PULSE_REQUIRE(a.obs && a.obs_offset >= 0 && a.num_samples >= 1 && a.num_samples <= 256 && a.num_height_points >= 0 &&
                      a.obs_stride >= a.obs_offset + 2 * a.num_samples + a.num_height_points, "pulse_traj_step: bad obs target");
        if (a.num_height_points > 0) {
            PULSE_REQUIRE(a.height_points && a.sensor_body >= 0 && a.sensor_body < a.num_bodies, "pulse_traj_step: height sensor needs its grid and a body");
            PULSE_REQUIRE(!a.heightsamples || (a.map_rows >= 2 && a.map_cols >= 2 && a.horizontal_scale > 0.f), "pulse_traj_step: bad height field");
            PULSE_REQUIRE(!a.use_center_height || !a.heightsamples || (a.center_points && a.num_center_points >= 1 && a.num_center_points <= 256),
                          "pulse_traj_step: use_center_height needs the centre grid (1 .. 256 points)");
        }
    }
    if (a.what & PULSE_TASK_REWARD) {
        PULSE_REQUIRE(a.rew, "pulse_traj_step: null rew");
        PULSE_REQUIRE(!a.dof_force || (a.dof_vel && a.num_dof >= 1), "pulse_traj_step: power term needs dof force and velocity");
        PULSE_REQUIRE(!a.power_reward || a.dof_force, "pulse_traj_step: power_reward needs dof_force");
    }
    if (a.what & PULSE_TASK_RESET) {
        PULSE_REQUIRE(a.reset && a.terminate, "pulse_traj_step: null reset outputs");
        if (a.enable_early_termination)
            PULSE_REQUIRE(a.contact_forces && (a.num_contact_ids == 0 || a.contact_body_ids) && (a.terrain_reset || a.termination_heights),
                          "pulse_traj_step: early termination inputs");
    }
    hipLaunchKernelGGL(traj_step_kernel, dim3((unsigned)count), dim3(256), 0, as_stream(s), a);
    return check_launch("pulse_traj_step");
}

int pulse_traj_generate(const pulse_traj_gen_args* args, pulse_stream_t s) {
    PULSE_REQUIRE(args != nullptr, "pulse_traj_generate: null args");
    const pulse_traj_gen_args& a = *args;
    if (a.num_envs == 0) return PULSE_OK;
    PULSE_REQUIRE(a.num_envs > 0 && a.num_verts >= 2, "pulse_traj_generate: bad sizes");
    PULSE_REQUIRE(a.rb && a.rb_env_stride >= 13 && a.verts && a.u_dtheta && a.u_sharp && a.sharp_mask && a.u_heading && a.u_dspeed && a.u_speed0,
                  "pulse_traj_generate: null pointer");
    hipLaunchKernelGGL(traj_generate_kernel, dim3((unsigned)((a.num_envs + 63) / 64)), dim3(64), 0, as_stream(s), a);
    return check_launch("pulse_traj_generate");
}
}
