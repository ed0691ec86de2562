/*
 * pulse_hip.h -- C ABI of the MI355X (gfx950) hot-path library for PULSE's
 * data-parallel RL training path.
 *
 * The reference (ZhengyiLuo/PULSE) is 100 % Python: its "kernels" are
 * @torch.jit.script functions and nn.Module calls, so there is no FFI in it to
 * bind against.  Every entry point below therefore replaces a Python-level
 * function of the reference; the file:line it stands for is cited on each
 * declaration (paths relative to the reference root).  INTEGRATION.md shows
 * the ctypes stub a maintainer would drop into phc/ to call them.
 *
 * Conventions
 *   - plain C, no torch types: raw DEVICE pointers + sizes + a hipStream_t
 *     passed as void* (0 = default stream).  All work is enqueued
 *     asynchronously on that stream; nothing synchronises.
 *   - caller owns every buffer; no allocation, no global state (except the
 *     thread-local last-error string).
 *   - return value: PULSE_OK (0) or a negative PULSE_ERR_* code; never throws.
 *   - quaternions are xyzw float32 (phc/utils/torch_utils.py:48).
 *   - "rb" is Isaac Gym's rigid-body record, 13 floats per body:
 *     pos 0:3, rot 3:7, lin vel 7:10, ang vel 10:13
 *     (phc/env/tasks/humanoid.py:215-222).
 */
#ifndef PULSE_HIP_H
#define PULSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PULSE_OK 0
#define PULSE_ERR_INVALID_ARG (-1)
#define PULSE_ERR_LAUNCH (-2)
#define PULSE_ERR_UNSUPPORTED (-3)

#define PULSE_ABI_VERSION 1

typedef void* pulse_stream_t; /* hipStream_t */

int pulse_abi_version(void);
/* Thread-local description of the last non-zero status returned on this thread. */
const char* pulse_last_error(void);

/* ------------------------------------------------------------------------- *
 * 1. Rotation algebra, one row per thread, m rows.
 *    isaacgym.torch_utils (3P; phc/utils/torch_utils.py:31) and
 *    phc/utils/torch_utils.py:45-240.
 * ------------------------------------------------------------------------- */
/* isaacgym quat_mul (factored 8-multiply form). a,b,out: (m,4) */
int pulse_quat_mul(const float* a, const float* b, float* out, int64_t m, pulse_stream_t s);
/* isaacgym quat_conjugate. */
int pulse_quat_conjugate(const float* a, float* out, int64_t m, pulse_stream_t s);
/* my_quat_rotate, torch_utils.py:45-55. q (m,4), v (m,3) -> out (m,3) */
int pulse_quat_rotate(const float* q, const float* v, float* out, int64_t m, pulse_stream_t s);
/* quat_to_angle_axis, torch_utils.py:57-78 -> angle (m), axis (m,3) */
int pulse_quat_to_angle_axis(const float* q, float* angle, float* axis, int64_t m, pulse_stream_t s);
/* quat_to_exp_map, torch_utils.py:81-97 -> (m,3) */
int pulse_quat_to_exp_map(const float* q, float* out, int64_t m, pulse_stream_t s);
/* quat_to_tan_norm (6-D rotation), torch_utils.py:100-113 -> (m,6) */
int pulse_quat_to_tan_norm(const float* q, float* out, int64_t m, pulse_stream_t s);
/* exp_map_to_quat, torch_utils.py:148-172. e (m,3) -> (m,4) */
int pulse_exp_map_to_quat(const float* e, float* out, int64_t m, pulse_stream_t s);
/* slerp, torch_utils.py:175-197. q0,q1 (m,4), t (m) -> (m,4) */
int pulse_slerp(const float* q0, const float* q1, const float* t, float* out, int64_t m, pulse_stream_t s);
/* calc_heading, torch_utils.py:200-212 -> (m) */
int pulse_calc_heading(const float* q, float* out, int64_t m, pulse_stream_t s);
/* calc_heading_quat / calc_heading_quat_inv, torch_utils.py:215-240 -> (m,4) */
int pulse_calc_heading_quat(const float* q, float* out, int64_t m, int inverse, pulse_stream_t s);

/* ------------------------------------------------------------------------- *
 * 2. Fused HumanoidIm post-physics step: reward -> reset -> next observation
 *    (phc/env/tasks/humanoid.py:1315-1331 order).
 * ------------------------------------------------------------------------- */
typedef struct pulse_reward_specs {
    /* phc/env/tasks/humanoid_im.py:55 (reward_specs) and :92 (power_coefficient) */
    float k_pos, k_rot, k_vel, k_ang_vel;
    float w_pos, w_rot, w_vel, w_ang_vel;
    float power_coef;
    int32_t power_reward; /* env_im.yaml:23 */
} pulse_reward_specs;

/* what pulse_im_step computes (bit mask in pulse_im_step_args.what) */
#define PULSE_IM_SELF_OBS 1u  /* compute_humanoid_observations_smpl_max, humanoid.py:1675-1731 */
#define PULSE_IM_TASK_OBS 2u  /* compute_imitation_observations_v6 / _v7, humanoid_im.py:1328-1413 */
#define PULSE_IM_REWARD   4u  /* compute_imitation_reward (+ power term), humanoid_im.py:853-919,1543-1574 */
#define PULSE_IM_RESET    8u  /* compute_humanoid_im_reset, humanoid_im.py:1119-1192,1600-1628 */

typedef struct pulse_im_step_args {
    /* ---- simulation state (read) ---- */
    const float* rb;          /* (num_envs, >=num_bodies, 13) */
    int64_t rb_env_stride;    /* floats between consecutive envs (>= num_bodies*13) */
    int32_t num_envs;
    int32_t num_bodies;       /* J, 24 for SMPL; <= 32 */
    /* optional subset selection (partial reset, humanoid_im.py:677-706 with env_ids):
       env_ids != NULL: process only env_ids[0..num_ids);  env_mask != NULL: skip
       envs whose mask byte is 0.  Both NULL: all envs. */
    const int64_t* env_ids;
    int32_t num_ids;
    const uint8_t* env_mask;

    /* ---- reference motion at motion time t (reward / reset) : (num_envs, J, 3|4) ---- */
    const float* ref_now_pos;
    const float* ref_now_rot;
    const float* ref_now_vel;
    const float* ref_now_ang;
    /* ---- reference motion at t+1 (task obs): (num_envs*time_steps, J, 3|4), env-major ---- */
    const float* ref_next_pos;
    const float* ref_next_rot;
    const float* ref_next_vel;
    const float* ref_next_ang;
    int32_t time_steps;       /* _num_traj_samples, 1 unless fut_tracks */

    /* ---- power reward inputs: (num_envs, num_dof) ---- */
    const float* dof_force;
    const float* dof_vel;
    int32_t num_dof;          /* 69 */

    /* ---- episode bookkeeping ---- */
    const int64_t* progress;      /* progress_buf (num_envs) */
    const uint8_t* pass_time;     /* bool (num_envs): time >= motion length (or episode cap) */
    const int64_t* cycle_counter; /* optional (num_envs): recovery mask, humanoid_im.py:1188-1190 */

    /* ---- body subsets ---- */
    const int32_t* track_ids; int32_t num_track;  /* _track_bodies_id */
    const int32_t* reset_ids; int32_t num_reset;  /* _reset_bodies_id */
    const float* term_dist;   /* (J) per-body termination distance, indexed by body id */
    int32_t reset_use_mean;   /* flags.im_eval && !strict_eval */
    int32_t full_body_reward; /* humanoid_im.py:37 */

    /* ---- options ---- */
    uint32_t what;            /* PULSE_IM_* mask */
    int32_t obs_version;      /* 6 or 7 */
    int32_t local_root_obs;   /* env_im.yaml:33 */
    int32_t root_height_obs;  /* env_im.yaml:34 */
    pulse_reward_specs specs;

    /* ---- outputs ---- */
    float* obs;               /* (num_envs, obs_stride): [self_obs | task_obs | zero pad] */
    int64_t obs_stride;       /* floats per row (>= obs_cols) */
    int32_t obs_cols;         /* columns written per row incl. zero padding (>= obs width) */
    float* rew;               /* (num_envs) */
    float* rew_raw;           /* (num_envs, 5 if power_reward else 4) */
    int64_t* reset;           /* (num_envs) */
    int64_t* terminate;       /* (num_envs) */
} pulse_im_step_args;

/* width of the self / task observation for the given options */
/* sizeof(pulse_im_step_args) as compiled, so a foreign-language binding can verify its mirror */
int pulse_sizeof_im_step_args(void);
int pulse_self_obs_width(int num_bodies, int root_height_obs);
int pulse_task_obs_width(int obs_version, int num_track, int time_steps);
int pulse_im_step(const pulse_im_step_args* args, pulse_stream_t s);

/* ------------------------------------------------------------------------- *
 * 3. GAE: CommonAgent.discount_values + returns, phc/learning/common_agent.py:493-505,
 *    :346-347.  Element (t, n) of every array lives at t*stride_t + n*stride_n.
 *    gamma_tau is the host-side double product gamma*tau rounded to float (as
 *    Python evaluates `self.gamma * self.tau * not_done`).
 * ------------------------------------------------------------------------- */
int pulse_gae(const float* rewards, const float* values, const float* next_values, const uint8_t* dones,
              int32_t horizon, int32_t num_envs, int64_t stride_t, int64_t stride_n,
              float gamma, float gamma_tau, float* advs, float* returns, pulse_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* PULSE_HIP_H */
