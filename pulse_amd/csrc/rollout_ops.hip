// Per-step rollout bookkeeping for gfx950 (play_steps, phc/learning/amp_agent.py:372-412 / common_agent.py:318-347)
// and the kinematic physics stand-in of the motion-library path.
//
// The reference issues ~45 tiny elementwise / reduction launches per rollout step for this (reward shaping, buffer
// writes, bootstrap masking, episode accumulators, two AverageMeter updates); at 4096 envs every one of them is pure
// launch latency (~5 us each, dependent).  pulse_rollout_record does all of it in one single-workgroup launch:
// 1024 threads stride over the envs, the three reductions the meters need (count / sum of finished-episode returns /
// sum of finished-episode lengths) go through wave shuffles + LDS in a fixed order (deterministic).
// HBM traffic is ~40 B per env: launch-latency bound by construction, which is the point.
// Compiled with -ffp-contract=off.
#include "common.h"
#include "rot_math.h"

namespace pulse {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// rl_games AverageMeter.update (torch_ext.py): windowed running mean over finished episodes
__device__ __forceinline__ void meter_update(float* st, float sum, float count, float max_size) {
    if (count <= 0.f) return;
    const float new_mean = sum / count;
    const float size = fminf(fmaxf(count, 0.f), max_size);
    const float old_size = fminf(max_size - size, st[1]);
    const float size_sum = old_size + size;
    st[0] = (st[0] * old_size + new_mean * size) / size_sum;
    st[1] = size_sum;
}

// One workgroup (meter_partials == NULL): the launch also updates the two AverageMeters.  Several workgroups (meter_partials given): every
// workgroup takes a contiguous range of envs and leaves its (count, return sum, length sum) of finished episodes in
// meter_partials[block * 4 ..]; pulse_rollout_meters applies the steps' meter updates in order after the rollout.  (One workgroup striding
// over 8192 envs is a chain of dependent loads: 31 us per rollout step; 32 workgroups: 6 us.)
__global__ void __launch_bounds__(1024) rollout_record_kernel(const pulse_rollout_record_args a) {
    __shared__ float red[3][16];
    float cnt = 0.f, sr = 0.f, sl = 0.f;
    float vs = 1.f, vm = 0.f;
    if (a.value_mean) {
        vs = sqrtf((float)a.value_var[0] + a.value_eps);       // running_mean_std.py:84-86 (unnorm): sqrt(var.float() + eps) * clamp(y) + mean.float()
        vm = (float)a.value_mean[0];
    }
    const int per = (a.num_envs + gridDim.x - 1) / gridDim.x;
    const int e0 = blockIdx.x * per, e1 = min(a.num_envs, e0 + per);
    for (int e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
        const float r = a.rewards[e];
        const bool done = a.dones[e] != 0;
        const long long o = (long long)e * a.env_stride;
        a.buf_rewards[o] = (a.reward_scale == 1.f && a.reward_shift == 0.f) ? r : (r + a.reward_shift) * a.reward_scale;
        a.buf_dones[o] = done ? 1 : 0;
        if (a.buf_terminate) a.buf_terminate[o] = a.terminate[e] != 0 ? 1 : 0;
        if (a.buf_next_values) {
            float v = a.value_raw[(long long)e * a.value_stride];
            if (a.value_mean) v = vs * fminf(fmaxf(v, -5.f), 5.f) + vm;
            a.buf_next_values[o] = v * (1.0f - (float)a.terminate[e]);
        }
        const float cr = a.current_rewards[e] + r;
        const float cl = a.current_lengths[e] + 1.f;
        if (done) { cnt += 1.f; sr += cr; sl += cl; }
        const float keep = 1.0f - (done ? 1.f : 0.f);
        a.current_rewards[e] = cr * keep;
        a.current_lengths[e] = cl * keep;
        a.done_mask[e] = done ? 1 : 0;
    }
    cnt = wave_sum_f(cnt); sr = wave_sum_f(sr); sl = wave_sum_f(sl);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = cnt; red[1][w] = sr; red[2][w] = sl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float c = 0.f, r = 0.f, l = 0.f;
        const int nw = blockDim.x >> 6;
        for (int i = 0; i < nw; ++i) { c += red[0][i]; r += red[1][i]; l += red[2][i]; }
        if (a.meter_partials) {
            float* o = a.meter_partials + 4 * blockIdx.x;
            o[0] = c; o[1] = r; o[2] = l; o[3] = 0.f;
        } else {
            meter_update(a.meter_rewards, r, c, a.meter_max_size);
            meter_update(a.meter_lengths, l, c, a.meter_max_size);
        }
    }
}

// the deferred AverageMeter updates of ``steps`` rollout steps, in step order (one thread: a few hundred adds)
__global__ void rollout_meters_kernel(const float* __restrict__ partials, int steps, int blocks, float* __restrict__ meter_rewards,
                                      float* __restrict__ meter_lengths, float max_size) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int t = 0; t < steps; ++t) {
        float c = 0.f, r = 0.f, l = 0.f;
        for (int b = 0; b < blocks; ++b) {
            const float* p = partials + ((long long)t * blocks + b) * 4;
            c += p[0]; r += p[1]; l += p[2];
        }
        meter_update(meter_rewards, r, c, max_size);
        meter_update(meter_lengths, l, c, max_size);
    }
}

// one thread per rigid body record / per dof
__global__ void __launch_bounds__(256) kinematic_sim_kernel(const float* __restrict__ trb, const float* __restrict__ nrb, float* __restrict__ rb,
                                                           long long bodies, const float* __restrict__ tdp, const float* __restrict__ ndp,
                                                           float* __restrict__ dp, const float* __restrict__ tdv, const float* __restrict__ ndv,
                                                           float* __restrict__ dv, const float* __restrict__ fsrc, float* __restrict__ f, long long dofs) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < bodies) {
        const float* t = trb + i * 13;
        const float* n = nrb + i * 13;
        float v[13];
#pragma unroll
        for (int k = 0; k < 13; ++k) v[k] = t[k] + n[k];
        const float nq = sqrtf(v[3] * v[3] + v[4] * v[4] + v[5] * v[5] + v[6] * v[6]);
#pragma unroll
        for (int k = 3; k < 7; ++k) v[k] = v[k] / nq;
        float* o = rb + i * 13;
#pragma unroll
        for (int k = 0; k < 13; ++k) o[k] = v[k];
    }
    if (i < dofs) {
        dp[i] = tdp[i] + ndp[i];
        dv[i] = tdv[i] + ndv[i];
        f[i] = fsrc[i];
    }
}

// action-dependent physics stand-in (see include/pulse_hip.h: pulse_pd_sim_args); one thread per (env, body)
__global__ void __launch_bounds__(256) pd_sim_kernel(const pulse_pd_sim_args a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int J = a.num_bodies, nd = 3 * (J - 1);
    if (i >= a.num_envs * J) return;
    const long long e = i / J;
    const int b = (int)(i - e * J);
    const float* t = a.target_rb + i * 13;
    float* o = a.rb + i * 13;
    if (b == 0) {
#pragma unroll
        for (int k = 0; k < 13; ++k) o[k] = t[k];
        return;
    }
    const long long d0 = e * nd + 3 * (b - 1);
    const bool rst = a.reset_mask && a.reset_mask[e];
    float er[3], ev[3], tq[3];
    const float h = a.dt / (float)a.substeps;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        er[k] = rst ? 0.0f : a.err[d0 + k];
        ev[k] = rst ? 0.0f : a.err_vel[d0 + k];
        tq[k] = 0.0f;
    }
    if (!rst) {
        for (int s = 0; s < a.substeps; ++s) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float te = a.sag[3 * (b - 1) + k] + a.action_scale * a.action[d0 + k];
                const float acc = a.kp * (te - er[k]) - a.kd * ev[k] + a.noise_acc[d0 + k];
                ev[k] = ev[k] + h * acc;
                er[k] = er[k] + h * ev[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float te = a.sag[3 * (b - 1) + k] + a.action_scale * a.action[d0 + k];
            tq[k] = a.kp * (te - er[k]) - a.kd * ev[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a.err[d0 + k] = er[k];
        a.err_vel[d0 + k] = ev[k];
        a.dof_pos[d0 + k] = a.target_dof_pos[d0 + k] + er[k];
        a.dof_vel[d0 + k] = a.target_dof_vel[d0 + k] + ev[k];
        a.dof_force[d0 + k] = tq[k];
    }
    const float* u = a.lever_dir + 3 * b;
    const V3 ce{er[1] * u[2] - er[2] * u[1], er[2] * u[0] - er[0] * u[2], er[0] * u[1] - er[1] * u[0]};
    const V3 cv{ev[1] * u[2] - ev[2] * u[1], ev[2] * u[0] - ev[0] * u[2], ev[0] * u[1] - ev[1] * u[0]};
    const Q4 q = qmul(exp_map_to_q(V3{er[0], er[1], er[2]}), Q4{t[3], t[4], t[5], t[6]});
    o[0] = t[0] + a.lever * ce.x; o[1] = t[1] + a.lever * ce.y; o[2] = t[2] + a.lever * ce.z;
    o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
    o[7] = t[7] + a.lever * cv.x; o[8] = t[8] + a.lever * cv.y; o[9] = t[9] + a.lever * cv.z;
    o[10] = t[10] + ev[0]; o[11] = t[11] + ev[1]; o[12] = t[12] + ev[2];
}

}  // namespace pulse

using namespace pulse;

extern "C" int pulse_sizeof_pd_sim_args(void) { return (int)sizeof(pulse_pd_sim_args); }

extern "C" int pulse_pd_sim_step(const pulse_pd_sim_args* args, pulse_stream_t s) {
    PULSE_REQUIRE(args != nullptr, "pulse_pd_sim_step: null args");
    const pulse_pd_sim_args& a = *args;
    PULSE_REQUIRE(a.num_envs >= 0 && a.num_bodies >= 2 && a.substeps >= 1 && a.dt > 0.f, "pulse_pd_sim_step: bad sizes");
    if (a.num_envs == 0) return PULSE_OK;
    PULSE_REQUIRE(a.target_rb && a.target_dof_pos && a.target_dof_vel && a.action && a.noise_acc && a.sag && a.lever_dir && a.err && a.err_vel &&
                      a.rb && a.dof_pos && a.dof_vel && a.dof_force, "pulse_pd_sim_step: null pointer");
    const long long n = a.num_envs * a.num_bodies;
    hipLaunchKernelGGL(pd_sim_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(s), a);
    return check_launch("pulse_pd_sim_step");
}

extern "C" int pulse_sizeof_rollout_record_args(void) { return (int)sizeof(pulse_rollout_record_args); }

extern "C" int pulse_rollout_record(const pulse_rollout_record_args* args, pulse_stream_t s) {
    PULSE_REQUIRE(args != nullptr, "pulse_rollout_record: null args");
    const pulse_rollout_record_args& a = *args;
    PULSE_REQUIRE(a.num_envs >= 0, "pulse_rollout_record: negative num_envs");
    if (a.num_envs == 0) return PULSE_OK;
    PULSE_REQUIRE(a.rewards && a.dones && a.buf_rewards && a.buf_dones && a.current_rewards && a.current_lengths && a.meter_rewards &&
                      a.meter_lengths && a.done_mask, "pulse_rollout_record: null pointer");
    PULSE_REQUIRE(a.buf_next_values == nullptr || (a.value_raw && a.terminate), "pulse_rollout_record: next_values needs value_raw and terminate");
    PULSE_REQUIRE(a.buf_terminate == nullptr || a.terminate, "pulse_rollout_record: buf_terminate needs terminate");
    PULSE_REQUIRE((a.value_mean == nullptr) == (a.value_var == nullptr), "pulse_rollout_record: value_mean / value_var go together");
    PULSE_REQUIRE(a.env_stride >= 1 && a.meter_max_size >= 1.f, "pulse_rollout_record: bad env_stride / meter_max_size");
    if (a.meter_partials) {
        PULSE_REQUIRE(a.meter_blocks >= 1 && a.meter_blocks <= 1024, "pulse_rollout_record: meter_blocks must be in [1, 1024]");
        hipLaunchKernelGGL(rollout_record_kernel, dim3((unsigned)a.meter_blocks), dim3(256), 0, as_stream(s), a);
    } else {
        hipLaunchKernelGGL(rollout_record_kernel, dim3(1), dim3(1024), 0, as_stream(s), a);
    }
    return check_launch("pulse_rollout_record");
}

extern "C" int pulse_rollout_meters(const float* partials, int32_t steps, int32_t blocks, float* meter_rewards, float* meter_lengths, float meter_max_size,
                                    pulse_stream_t s) {
    PULSE_REQUIRE(steps >= 0 && blocks >= 1, "pulse_rollout_meters: bad sizes");
    if (steps == 0) return PULSE_OK;
    PULSE_REQUIRE(partials && meter_rewards && meter_lengths && meter_max_size >= 1.f, "pulse_rollout_meters: bad arguments");
    hipLaunchKernelGGL(rollout_meters_kernel, dim3(1), dim3(64), 0, as_stream(s), partials, steps, blocks, meter_rewards, meter_lengths, meter_max_size);
    return check_launch("pulse_rollout_meters");
}

extern "C" int pulse_kinematic_sim_step(const float* target_rb, const float* noise_rb, float* rb, int64_t num_envs, int32_t num_bodies,
                                        const float* target_dof_pos, const float* noise_dof_pos, float* dof_pos,
                                        const float* target_dof_vel, const float* noise_dof_vel, float* dof_vel,
                                        const float* force_src, float* dof_force, int32_t num_dof, pulse_stream_t s) {
    PULSE_REQUIRE(num_envs >= 0 && num_bodies >= 1 && num_dof >= 1, "pulse_kinematic_sim_step: bad sizes");
    if (num_envs == 0) return PULSE_OK;
    PULSE_REQUIRE(target_rb && noise_rb && rb && target_dof_pos && noise_dof_pos && dof_pos && target_dof_vel && noise_dof_vel && dof_vel &&
                      force_src && dof_force, "pulse_kinematic_sim_step: null pointer");
    const long long bodies = (long long)num_envs * num_bodies, dofs = (long long)num_envs * num_dof;
    const long long n = bodies > dofs ? bodies : dofs;
    hipLaunchKernelGGL(kinematic_sim_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(s), target_rb, noise_rb, rb, bodies,
                       target_dof_pos, noise_dof_pos, dof_pos, target_dof_vel, noise_dof_vel, dof_vel, force_src, dof_force, dofs);
    return check_launch("pulse_kinematic_sim_step");
}
