// Downstream-task observation / reward / reset kernels for gfx950: the tasks a frozen PULSE
// decoder is trained on (HumanoidSpeedZ / HumanoidReachZ / HumanoidStrikeZ in the README's commands).
//
// Replaces the reference's TorchScript functions
//   compute_speed_observations / compute_speed_reward      phc/env/tasks/humanoid_speed.py:310-343 (+ power term :211-218)
//   compute_location_observations / compute_reach_reward   phc/env/tasks/humanoid_reach.py:224-250
//   compute_strike_observations / compute_strike_reward    phc/env/tasks/humanoid_strike.py:270-327
//   compute_humanoid_reset (+ the strike variant)          phc/env/tasks/humanoid.py:1572-1608, humanoid_strike.py:330-380
// One thread per environment: the per-env state of these tasks is a root record, a target and a handful of scalars
// (< 500 B with the 24 x 3 contact forces), so the work is launch-latency bound; what matters is that one launch replaces
// the ~40 elementwise launches of the eager form and that every output goes straight into the caller's buffers (the task
// observation lands at its column offset of the GEMM-ready observation row).  -ffp-contract=off, reference op order.
#include "common.h"
#include "rot_math.h"

namespace pulse {

__device__ __forceinline__ int task_obs_width(int task) {
    return task == PULSE_TASK_SPEED ? 3 : task == PULSE_TASK_REACH ? 3 : 15;
}

// compute_humanoid_reset's fall test: contact on a non-foot body AND some non-foot body below its termination height
__device__ __forceinline__ bool has_fallen(const pulse_task_step_args& a, const float* rb, const float* cf, bool* nonstrike_contact) {
    bool fall_contact = false, fall_height = false, nonstrike = false;
    for (int b = 0; b < a.num_bodies; ++b) {
        bool is_contact_body = false, is_strike_body = false;
        for (int k = 0; k < a.num_contact_ids; ++k) is_contact_body |= a.contact_body_ids[k] == b;
        for (int k = 0; k < a.num_strike; ++k) is_strike_body |= a.strike_body_ids[k] == b;
        if (is_contact_body) continue;                           // masked_contact_buf[:, contact_body_ids, :] = 0
        const float* f = cf + 3 * b;
        const float ax = fabsf(f[0]), ay = fabsf(f[1]), az = fabsf(f[2]);
        fall_contact |= (ax > 0.1f) || (ay > 0.1f) || (az > 0.1f);
        fall_height |= rb[13 * b + 2] < a.termination_heights[b];
        if (!is_strike_body) nonstrike |= (ax > 50.0f) || (ay > 50.0f) || (az > 50.0f);
    }
    *nonstrike_contact = nonstrike;
    return fall_contact && fall_height;
}

__global__ void __launch_bounds__(256) task_step_kernel(const pulse_task_step_args a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int count = a.env_ids ? a.num_ids : a.num_envs;
    if (i >= count) return;
    const int64_t e = a.env_ids ? a.env_ids[i] : (int64_t)i;
    if (a.env_mask && a.env_mask[e] == 0) return;
    const float* rb = a.rb + e * a.rb_env_stride;
    const V3 root_p{rb[0], rb[1], rb[2]};
    const Q4 root_q{rb[3], rb[4], rb[5], rb[6]};

    if (a.what & PULSE_TASK_OBS) {
        float* o = a.obs + e * a.obs_stride + a.obs_offset;
        const Q4 hinv = heading_quat(root_q, true);
        if (a.task == PULSE_TASK_SPEED) {
            const V3 l = qrot(hinv, V3{1.0f, 0.0f, 0.0f});                      // tar_dir3d = [1, 0, 0]
            o[0] = l.x; o[1] = l.y; o[2] = a.tar_speed[e];
        } else if (a.task == PULSE_TASK_REACH) {
            const float* t = a.tar_pos + 3 * e;
            const V3 l = qrot(hinv, V3{t[0] - root_p.x, t[1] - root_p.y, t[2] - root_p.z});
            o[0] = l.x; o[1] = l.y; o[2] = l.z;
        } else {
            const float* t = a.tar_states + 13 * e;
            const V3 lp = qrot(hinv, V3{t[0] - root_p.x, t[1] - root_p.y, t[2]});   // local_tar_pos[..., -1] = tar_pos[..., -1]
            float tn[6];
            q_to_tan_norm(qmul(hinv, Q4{t[3], t[4], t[5], t[6]}), tn);
            const V3 lv = qrot(hinv, V3{t[7], t[8], t[9]});
            const V3 lw = qrot(hinv, V3{t[10], t[11], t[12]});
            o[0] = lp.x; o[1] = lp.y; o[2] = lp.z;
#pragma unroll
            for (int k = 0; k < 6; ++k) o[3 + k] = tn[k];
            o[9] = lv.x; o[10] = lv.y; o[11] = lv.z;
            o[12] = lw.x; o[13] = lw.y; o[14] = lw.z;
        }
    }

    if (a.what & PULSE_TASK_REWARD) {
        const float* pp = a.prev_root_pos ? a.prev_root_pos + 3 * e : rb;
        float rew;
        if (a.task == PULSE_TASK_SPEED) {
            const float vx = (root_p.x - pp[0]) / a.dt, vy = (root_p.y - pp[1]) / a.dt;
            const float err = a.tar_speed[e] - vx;
            rew = expf(-0.25f * (err * err + 0.1f * vy * vy));
        } else if (a.task == PULSE_TASK_REACH) {
            const float* t = a.tar_pos + 3 * e;
            const float* r = rb + 13 * a.reach_body_id;
            const float dx = t[0] - r[0], dy = t[1] - r[1], dz = t[2] - r[2];
            rew = expf(-4.0f * (dx * dx + dy * dy + dz * dz));
        } else {
            const float* t = a.tar_states + 13 * e;
            // isaacgym quat_rotate(tar_rot, up): a = v (2 w^2 - 1), b = 2 w (q x v), c = 2 q (q . v), with v = [0, 0, 1]
            const float qx = t[3], qy = t[4], qz = t[5], qw = t[6];
            const float s = 2.0f * qw * qw - 1.0f;
            const V3 av{0.0f * s, 0.0f * s, 1.0f * s};
            const V3 bv{(qy * 1.0f - qz * 0.0f) * qw * 2.0f, (qz * 0.0f - qx * 1.0f) * qw * 2.0f, (qx * 0.0f - qy * 0.0f) * qw * 2.0f};
            const float dot = qx * 0.0f + qy * 0.0f + qz * 1.0f;
            const V3 cv{qx * dot * 2.0f, qy * dot * 2.0f, qz * dot * 2.0f};
            const float rz = av.z + bv.z + cv.z, rx = av.x + bv.x + cv.x, ry = av.y + bv.y + cv.y;
            const float rot_err = 0.0f * rx + 0.0f * ry + 1.0f * rz;
            const float rot_r = fmaxf(1.0f - rot_err, 0.0f);
            float dx = t[0] - root_p.x, dy = t[1] - root_p.y;
            const float nrm = fmaxf(sqrtf(dx * dx + dy * dy), 1e-12f);          // F.normalize eps
            dx = dx / nrm; dy = dy / nrm;
            const float vx = (root_p.x - pp[0]) / a.dt, vy = (root_p.y - pp[1]) / a.dt;
            const float dir_speed = dx * vx + dy * vy;
            const float verr = fmaxf(1.0f - dir_speed, 0.0f);
            float vel_r = expf(-4.0f * (verr * verr));
            if (dir_speed <= 0.0f) vel_r = 0.0f;
            rew = 0.6f * rot_r + 0.4f * vel_r;
            if (rot_err < 0.2f) rew = 1.0f;
        }
        float* raw = a.rew_raw ? a.rew_raw + e * a.rew_raw_width : nullptr;
        if (raw) raw[0] = rew;
        if (a.power_reward) {                                                   // humanoid_speed.py:211-218
            float pw = 0.0f;
            const float* f = a.dof_force + e * a.num_dof;
            const float* v = a.dof_vel + e * a.num_dof;
            for (int d = 0; d < a.num_dof; ++d) pw += fabsf(f[d] * v[d]);
            float p = -a.power_coef * pw;
            if (a.progress[e] <= 3) p = 0.0f;
            rew += p;
            if (raw && a.rew_raw_width > 1) raw[1] = p;
        }
        a.rew[e] = rew;
    }

    if (a.what & PULSE_TASK_RESET) {
        const long long prog = a.progress[e];
        int64_t term = 0;
        if (a.enable_early_termination) {
            bool nonstrike = false;
            bool failed = has_fallen(a, rb, a.contact_forces + e * (3 * a.num_bodies), &nonstrike);
            if (a.task == PULSE_TASK_STRIKE) {
                const float* tf = a.tar_contact_forces + 3 * e;
                const bool tar_contact = (fabsf(tf[0]) > 50.0f) || (fabsf(tf[1]) > 50.0f);
                failed = failed || (tar_contact && nonstrike);
            }
            term = (failed && prog > 1) ? 1 : 0;
        }
        a.reset[e] = ((float)prog >= a.max_episode_length - 1.0f) ? 1 : term;
        a.terminate[e] = term;
    }
}

}  // namespace pulse

using namespace pulse;

extern "C" {

int pulse_sizeof_task_step_args(void) { return (int)sizeof(pulse_task_step_args); }
int pulse_task_obs_size(int task) { return task == PULSE_TASK_SPEED ? 3 : task == PULSE_TASK_REACH ? 3 : task == PULSE_TASK_STRIKE ? 15 : -1; }

int pulse_task_step(const pulse_task_step_args* args, pulse_stream_t s) {
    PULSE_REQUIRE(args != nullptr, "pulse_task_step: null args");
    const pulse_task_step_args& a = *args;
    PULSE_REQUIRE(a.task == PULSE_TASK_SPEED || a.task == PULSE_TASK_REACH || a.task == PULSE_TASK_STRIKE, "pulse_task_step: unknown task %d", a.task);
    PULSE_REQUIRE(a.num_envs >= 0, "pulse_task_step: negative num_envs");
    const int count = a.env_ids ? a.num_ids : a.num_envs;
    if (count == 0 || a.what == 0) return PULSE_OK;
    PULSE_REQUIRE(a.rb != nullptr && a.num_bodies >= 1 && a.rb_env_stride >= 13LL * a.num_bodies, "pulse_task_step: bad rigid-body state");
    if (a.what & PULSE_TASK_OBS) {
        PULSE_REQUIRE(a.obs != nullptr && a.obs_offset >= 0 && a.obs_stride >= a.obs_offset + pulse_task_obs_size(a.task), "pulse_task_step: bad obs target");
    }
    if (a.what & (PULSE_TASK_OBS | PULSE_TASK_REWARD)) {
        if (a.task == PULSE_TASK_SPEED) PULSE_REQUIRE(a.tar_speed != nullptr, "pulse_task_step: speed task needs tar_speed");
        if (a.task == PULSE_TASK_REACH) PULSE_REQUIRE(a.tar_pos != nullptr && a.reach_body_id >= 0 && a.reach_body_id < a.num_bodies, "pulse_task_step: reach task needs tar_pos / reach body");
        if (a.task == PULSE_TASK_STRIKE) PULSE_REQUIRE(a.tar_states != nullptr, "pulse_task_step: strike task needs tar_states");
    }
    if (a.what & PULSE_TASK_REWARD) {
        PULSE_REQUIRE(a.rew != nullptr && a.dt > 0.f, "pulse_task_step: reward needs rew and dt");
        PULSE_REQUIRE(a.task == PULSE_TASK_REACH || a.prev_root_pos != nullptr, "pulse_task_step: reward needs prev_root_pos");
        if (a.power_reward) PULSE_REQUIRE(a.dof_force && a.dof_vel && a.num_dof >= 1 && a.progress, "pulse_task_step: power reward inputs");
        PULSE_REQUIRE(a.rew_raw == nullptr || a.rew_raw_width >= (a.power_reward ? 2 : 1), "pulse_task_step: rew_raw_width too small");
    }
    if (a.what & PULSE_TASK_RESET) {
        PULSE_REQUIRE(a.reset && a.terminate && a.progress, "pulse_task_step: null reset inputs / outputs");
        if (a.enable_early_termination) {
            PULSE_REQUIRE(a.contact_forces && a.termination_heights && (a.num_contact_ids == 0 || a.contact_body_ids), "pulse_task_step: early termination inputs");
            if (a.task == PULSE_TASK_STRIKE)
                PULSE_REQUIRE(a.tar_contact_forces && (a.num_strike == 0 || a.strike_body_ids), "pulse_task_step: strike reset inputs");
        }
    }
    hipLaunchKernelGGL(task_step_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, as_stream(s), a);
    return check_launch("pulse_task_step");
}
}
