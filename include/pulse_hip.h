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

#define PULSE_ABI_VERSION 29

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
/* Packed reference-motion library (layout and semantics: section 2b). */
typedef struct pulse_motion_tables {
    const float* frames;              /* (total_frames, frame_stride) packed records */
    int64_t frame_stride;             /* floats per record, multiple of 4 */
    int64_t total_frames;
    int32_t num_bodies;               /* J <= 32 */
    int32_t off_gts, off_grs, off_lrs, off_gvs, off_gavs, off_dvs;   /* float offsets of the fields inside a record */
    const float* motion_lengths;      /* (num_motions) seconds  (_motion_lengths) */
    const float* motion_dt;           /* (num_motions)          (_motion_dt) */
    const int64_t* motion_num_frames; /* (num_motions)          (_motion_num_frames) */
    const int64_t* length_starts;     /* (num_motions) first record of each motion (:311-314) */
    int32_t num_motions;
} pulse_motion_tables;

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
#define PULSE_IM_DEBUG_POISON_LDS 0x80000000u /* debug: the kernel pre-fills its LDS with NaN (uninitialised-read detector) */

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
    int32_t obs_version;      /* task observation: 1, 2, 3, 6, 7, 8 (time_steps 1), 9  (humanoid_im.py:1222-1540) */
    int32_t local_root_obs;   /* env_im.yaml:33 */
    int32_t root_height_obs;  /* env_im.yaml:34 */
    pulse_reward_specs specs;
    int32_t upright_start;    /* robot/smpl_humanoid.yaml:7 has_upright_start; 0: heading from remove_base_rot(root), humanoid.py:1616-1620 */
    int32_t enable_early_termination; /* env_im.yaml enableEarlyTermination (compute_humanoid_im_reset's flag) */
    /* self observation variant (humanoid.py:1675-1849): 1 = _smpl_max; 3 = _smpl_max_v3 (force_sensor rows of force_sensor_width
       floats appended); 2 = _smpl_max_v2: ``rb`` holds hist_steps records per env, oldest first ((num_envs, hist_steps, J, 13),
       rb_env_stride >= hist_steps * J * 13), every step expressed in the heading frame of the newest root; reward / reset /
       task observations use the newest record. */
    int32_t self_obs_version; int32_t hist_steps;
    const float* force_sensor; int32_t force_sensor_width;
    /* obs_version 2 only (time_steps 1): simulated and reference dof positions, (num_envs, 3 (J-1)); the observation takes the
       joints of the tracked bodies without the root (humanoid_im.py:755-758).  With use_motion the reference is blended in-kernel. */
    const float* dof_pos; const float* ref_next_dof_pos;

    /* ---- outputs ---- */
    float* obs;               /* (num_envs, obs_stride): [self_obs | task_obs | zero pad] */
    int64_t obs_stride;       /* floats per row (>= obs_cols) */
    int32_t obs_cols;         /* columns written per row incl. zero padding (>= obs width) */
    float* rew;               /* (num_envs) */
    float* rew_raw;           /* (num_envs, 5 if power_reward else 4) */
    int64_t* reset;           /* (num_envs) */
    int64_t* terminate;       /* (num_envs) */

    /* ---- optional: episode clock advanced in-kernel (post_physics_step: progress_buf += 1, humanoid.py:1316;
       _compute_reset: pass_time, humanoid_im.py:1120-1123,1148).  progress_rw != NULL: every processed env uses
       p = progress_rw[e] + progress_inc as its progress (``progress`` is ignored) and p is written back.
       clock_motion_len != NULL: pass_time = cycle_motion ? p >= max_episode_length - 1
                                                          : p * clock_dt + start_times[e] + start_offsets[e] >= clock_motion_len[e]
       (``pass_time`` is ignored; the flag is also stored to pass_time_out when given). */
    int64_t* progress_rw; int32_t progress_inc;
    float clock_dt;           /* control step, also the time base of the in-kernel reference below */
    const float* clock_start_times; const float* clock_start_offsets; const float* clock_motion_len;
    int32_t cycle_motion; int32_t max_episode_length;
    uint8_t* pass_time_out;

    /* ---- optional: reference motion evaluated in-kernel from the packed library (use_motion != 0) instead of
       ref_now_* / ref_next_*: two frame records per (env, time) are blended in the staging phase
       (get_motion_state, motion_lib_base.py:434-517) at t = p * clock_dt + start + offset (reward / reset) and
       (p + 1) * clock_dt + k * traj_dt + start + offset, k < time_steps (task obs; humanoid_im.py:723-735).
       track_*: optional outputs of the k = 0 next-step reference as simulator-layout records / dofs. */
    int32_t use_motion;
    pulse_motion_tables motion;
    const int64_t* motion_ids;    /* (num_envs) _sampled_motion_ids */
    const float* motion_offset;   /* (num_envs, 3) _global_offset or NULL */
    float traj_dt;
    float* track_rb; int64_t track_rb_stride;   /* (num_envs, J, 13) */
    float* track_dof_pos; float* track_dof_vel; /* (num_envs, (J-1)*3) */

    /* ---- optional: shape / limb-weight rows appended to the self observation (humanoid.py:1724-1728, 1843-1847: has_smpl_params /
       has_limb_weight_params; self_obs_version 1 and 3 -- the reference's _v2 raises for them): (num_envs, width) each, row stride given */
    const float* smpl_params; int32_t smpl_params_width; int64_t smpl_params_stride;
    const float* limb_weights; int32_t limb_weights_width; int64_t limb_weights_stride;
    /* ---- optional: HumanoidImGetup._compute_reset (humanoid_im_getup.py:203-210) folded into the reset stage: envs whose
       recovery_counter is > 0 neither reset nor terminate and their progress does not advance (the value written back to
       progress_rw is p - 1, and the observation stage runs on that clock, as in the reference where _compute_observations follows
       the decrement, humanoid.py:1325-1328).  Needs progress_rw; only read when ``what`` includes PULSE_IM_RESET. */
    const int32_t* recovery_counter;
    /* ---- optional: ``zero_out_far`` (humanoid.py:311-329; True in phc_kp_pnn_iccv.yaml:36, phc_shape_pnn_iccv.yaml:39, phc_kp_mcp_iccv.yaml:34,
       phc_shape_mcp_iccv.yaml:36, env_im_getup_mcp.yaml:27).  time_steps must be 1 (the reference's blocks broadcast (N, 3) against (N T, 3)).
       Task observation (HumanoidIm._compute_task_obs, humanoid_im.py:763-777 for obs_version 6 / 8 / 9, :814-826 for 7):
         d = |root_pos - ref_next_pos[track_ids[0]]|;  point_goal[e] = d;
         d > close_distance: the reference of tracked bodies 1.. (pos, rot) and of every tracked body (vel, ang vel) := the simulated state;
         d > far_distance:   ref pos of tracked body 0 := (ref - cur) / d * far_distance + cur   (a direction);
       Reward (HumanoidIm._compute_reward, :870-887; compute_point_goal_reward :1577-1582), read BEFORE the observation stage overwrites it:
         g = min(point_goal[e] - |root_pos - ref_now_pos[0]|, 1/3) * 9;  outside 0.25 m: reward = g, raw = (g, 0, 0, 0);
         inside: reward = g + 0.5 * full-body imitation reward, raw = (g, 0, 0, 0) + 0.5 * imitation raw;  then the power term as usual.
       The reset stage is unchanged (:1158-1176 calls compute_humanoid_im_reset on the un-masked reference). */
    int32_t zero_out_far; float close_distance; float far_distance;
    float* point_goal;        /* (num_envs) HumanoidIm._point_goal, read by the reward stage and written by the task-observation stage */
    /* ---- optional (v27): ``occl_training`` (humanoid.py:323-324; HumanoidIm.random_occlu_idx, humanoid_im.py:85, 1046-1058).  occl_bits[e] bit j set:
       tracked body j of env e is occluded -- the task observation sees the simulated state as that body's reference (pos, rot, vel, ang vel for
       obs_version 6 / 8 / 9, :778-784; pos / rot for 7, :827-831; applied after the zero_out_far masking), and, when occl_reset != 0
       (HumanoidIm._compute_reset's else-branch, :1178-1183, which indexes the mask by BODY id: tracked bodies must be all bodies in order),
       an occluded reset body never counts as fallen. */
    const uint32_t* occl_bits; int32_t occl_reset;
    /* ---- optional (v29): a second destination for the observation rows (obs_copy[e * obs_copy_stride + c], same columns as obs).  The rollout
       records every observation twice -- as the policy's next input and as ``next_obses`` of the step that produced it (a2c_common.play_steps:
       experience_buffer.update_data('next_obses', n, ...)) --; the step kernel holds the finished row in LDS, so the second copy costs a store,
       not a launch. */
    float* obs_copy; int64_t obs_copy_stride;
} pulse_im_step_args;

/* sizeof(pulse_im_step_args) as compiled, so a foreign-language binding can verify its mirror */
int pulse_sizeof_im_step_args(void);
/* width of the self / task observation for the given options (self_obs_version 1; _ex covers versions 2 / 3 and appended rows:
   its last argument is the TOTAL width of the rows appended to the self observation -- force sensors (version 3) + shape + limb-weight
   parameters; ignored for version 2) */
int pulse_self_obs_width(int num_bodies, int root_height_obs);
int pulse_self_obs_width_ex(int num_bodies, int root_height_obs, int self_obs_version, int hist_steps, int force_sensor_width);
int pulse_task_obs_width(int obs_version, int num_track, int time_steps);
int pulse_im_step(const pulse_im_step_args* args, pulse_stream_t s);

/* ------------------------------------------------------------------------- *
 * 2a'. Downstream tasks on a frozen PULSE decoder (speed / reach / strike):
 *    compute_speed_observations / compute_speed_reward      phc/env/tasks/humanoid_speed.py:310-343 (+ power term :211-218)
 *    compute_location_observations / compute_reach_reward   phc/env/tasks/humanoid_reach.py:224-250
 *    compute_strike_observations / compute_strike_reward    phc/env/tasks/humanoid_strike.py:270-327
 *    compute_humanoid_reset (+ strike variant)              phc/env/tasks/humanoid.py:1572-1608, humanoid_strike.py:330-380
 * ------------------------------------------------------------------------- */
#define PULSE_TASK_SPEED  1
#define PULSE_TASK_REACH  2
#define PULSE_TASK_STRIKE 3
#define PULSE_TASK_OBS    1u
#define PULSE_TASK_REWARD 2u
#define PULSE_TASK_RESET  4u

typedef struct pulse_task_step_args {
    int32_t task;             /* PULSE_TASK_SPEED | _REACH | _STRIKE */
    uint32_t what;            /* PULSE_TASK_OBS | _REWARD | _RESET mask */
    int32_t num_envs;
    const int64_t* env_ids; int32_t num_ids; const uint8_t* env_mask;   /* subset selection as in pulse_im_step */
    const float* rb; int64_t rb_env_stride; int32_t num_bodies;        /* (num_envs, J, 13); record 0 is _humanoid_root_states */
    const float* prev_root_pos;       /* (num_envs, 3)  _prev_root_pos (pre_physics_step) */
    float dt;
    const float* tar_speed;           /* speed: (num_envs) */
    const float* tar_pos;             /* reach: (num_envs, 3) */
    int32_t reach_body_id;            /* reach: _reach_body_id */
    const float* tar_states;          /* strike: (num_envs, 13) target object root state */
    const float* tar_contact_forces;  /* strike: (num_envs, 3) */
    const int32_t* strike_body_ids; int32_t num_strike;
    /* compute_humanoid_reset */
    const float* contact_forces;      /* (num_envs, J, 3) */
    const int32_t* contact_body_ids; int32_t num_contact_ids;
    const float* termination_heights; /* (J) */
    const int64_t* progress; float max_episode_length; int32_t enable_early_termination;
    /* power term of the reward */
    const float* dof_force; const float* dof_vel; int32_t num_dof; float power_coef; int32_t power_reward;
    /* outputs */
    float* obs; int64_t obs_stride; int32_t obs_offset;   /* task obs -> row e, columns [obs_offset, obs_offset + width) */
    float* rew; float* rew_raw; int32_t rew_raw_width;    /* rew_raw optional: [task reward, power term] */
    int64_t* reset; int64_t* terminate;
} pulse_task_step_args;

int pulse_sizeof_task_step_args(void);
int pulse_task_obs_size(int task);    /* 3 / 3 / 15 */
int pulse_task_step(const pulse_task_step_args* args, pulse_stream_t s);

/* AMP per-frame observation: build_amp_observations_smpl (phc/env/tasks/humanoid_amp.py:925-969) +
 * dof_to_obs_smpl (phc/env/tasks/humanoid.py:1436-1446). */
typedef struct pulse_amp_obs_args {
    const float* rb; int64_t rb_env_stride;  /* (num_envs, bodies, 13); root = body 0 */
    const float* dof_pos; const float* dof_vel; int32_t num_dof;   /* (num_envs, num_dof) exp-map dofs, 3 per joint */
    int32_t num_envs;
    const int64_t* env_ids; int32_t num_ids; const uint8_t* env_mask;   /* optional subset (as pulse_im_step) */
    const int32_t* joint_ids; int32_t num_joints;     /* dof joints entering the obs (NULL = 0..num_joints-1); dof_subset */
    uint32_t zero_joint_mask;                          /* bit j: joint j's dof pos / vel read as zero (humanoid_amp.py:636-639) */
    const int32_t* key_body_ids; int32_t num_key_bodies;
    int32_t local_root_obs, root_height_obs;
    float* out; int64_t out_stride;                    /* (num_envs, out_stride), first W columns written */
    /* history mode (hist_steps > 1): ``out`` is slot 0 of the env's (hist_steps, W) window (HumanoidAMP._amp_obs_buf, humanoid_amp.py:296-314,
       out_stride >= hist_steps * W): the launch moves frames 0 .. S-2 to 1 .. S-1 (_update_hist_amp_obs, :622-631), writes the current
       frame into slot 0 (HumanoidAMP.post_physics_step, :194-210) and, if window_out != NULL, copies the finished window to
       window_out + e * window_stride (the experience-buffer slot, amp_agent.py:377). */
    int32_t hist_steps; float* window_out; int64_t window_stride;
} pulse_amp_obs_args;
int pulse_sizeof_amp_obs_args(void);
int pulse_amp_obs_width(int num_joints, int num_key_bodies, int root_height_obs);
int pulse_amp_obs(const pulse_amp_obs_args* args, pulse_stream_t s);

/* _init_amp_obs_ref (phc/env/tasks/humanoid_amp.py:531-563) for the masked envs: history slot k + 1 (k = 0 .. hist_steps - 2) of env e :=
 * build_amp_observations_smpl of the env's motion at start_times[e] - dt * (k + 1) (MotionLibBase.get_motion_state, motion_lib_base.py:
 * 434-517, no root offset; no zeroed joints).  hist[e * env_stride + slot * step_stride + c]. */
typedef struct pulse_amp_hist_args {
    pulse_motion_tables tab; const int64_t* motion_ids; const float* start_times; float dt;
    int32_t num_envs; const uint8_t* env_mask; int32_t hist_steps;
    const int32_t* joint_ids; int32_t num_joints; const int32_t* key_body_ids; int32_t num_key_bodies;
    int32_t local_root_obs, root_height_obs;
    float* hist; int64_t env_stride, step_stride;
} pulse_amp_hist_args;
int pulse_sizeof_amp_hist_args(void);
int pulse_amp_hist_init(const pulse_amp_hist_args* args, pulse_stream_t s);

/* ------------------------------------------------------------------------- *
 * 2b. Reference-motion query: MotionLibBase.get_motion_state / get_root_pos_smpl /
 *     _calc_frame_blend (phc/utils/motion_lib_base.py:434-565).
 *
 *     Device data format.  The reference keeps six flat per-frame tables (gts, grs, lrs,
 *     gvs, gavs, dvs; :297-304) and gathers two rows from each.  Here ONE packed table
 *     holds a fixed-size record per frame with the six fields at the float offsets off_*
 *     (pulse_amd packs [grs J*4 | lrs J*4 | gts J*3 | gvs J*3 | gavs J*3 | dvs (J-1)*3 | pad]),
 *     so a query touches two contiguous records (SMPL: 477 floats, pitch 480 = 1920 B)
 *     instead of twelve scattered rows.  Frame f of motion m is record length_starts[m]+f.
 * ------------------------------------------------------------------------- */
/* pulse_motion_tables: declared in section 2 (the fused env step can evaluate the reference in-kernel too). */

typedef struct pulse_motion_state_args {
    pulse_motion_tables tab;
    int64_t n;                        /* queries */
    const int64_t* motion_ids;        /* (n) */
    /* query time: motion_times (n) if non-NULL; otherwise the episode clock is evaluated in-kernel
       (humanoid_im.py:723-731, 859): query i belongs to env e = i / time_steps, sample k = i % time_steps,
           t = (progress[e] + step_shift) * dt + k * traj_dt + start_times[e] + start_offsets[e],
       and motion_ids / offset are then indexed by e as well (n = num_envs * time_steps). */
    const float* motion_times;
    const int64_t* progress; int32_t step_shift; float dt;
    const float* start_times; const float* start_offsets;
    int32_t time_steps; float traj_dt;   /* _num_traj_samples (>= 1) and _traj_sample_timestep; clock mode only */
    const float* offset;              /* optional (n,3) added to every body position (_global_offset) */
    int32_t root_only;                /* get_root_pos_smpl: only root_pos is produced */
    /* outputs, contiguous; any pointer may be NULL to skip that field */
    float* rg_pos;                    /* (n, J, 3) */
    float* rb_rot;                    /* (n, J, 4) */
    float* body_vel;                  /* (n, J, 3) */
    float* body_ang_vel;              /* (n, J, 3) */
    float* dof_pos;                   /* (n, (J-1)*3) exp-map of the slerped local rotations (:562-565) */
    float* dof_vel;                   /* (n, (J-1)*3) */
    float* root_pos;                  /* (n, 3)  (root_only mode) */
    float* rb_records;                /* optional (n, J, 13) [pos | rot | vel | ang vel] records, the simulator's rigid-body */
    int64_t rb_query_stride;          /*   layout (humanoid.py:219-222): reference-state init writes them as-is; floats per query */
    int64_t* frame_idx0; int64_t* frame_idx1; float* blend;   /* optional (n): _calc_frame_blend results, idx RELATIVE to the motion */
    /* reset mode (reset_mask != NULL; per env, n = num_envs): reference-state init of the masked envs in one launch
       (_reset_envs -> _sample_ref_state, humanoid_im.py:966-986).  For env e with reset_mask[e] != 0:
           start = reset_phase ? reset_phase[e] * motion_length : 0      (MotionLibBase.sample_time with phase ~ U[0,1))
           reset_start_times[e] = start; reset_progress[e] = 0; reset_clear0/1[e] = 0   (reset_buf, _terminate_buf)
           outputs at row e := state at time step_shift * dt + start + start_offsets[e]
       rows of unmasked envs are left untouched (so the outputs can BE the simulator's state tensors). */
    const uint8_t* reset_mask; const float* reset_phase;
    float* reset_start_times; int64_t* reset_progress; int64_t* reset_clear0; int64_t* reset_clear1;
    /* reset_time_interval != 0: start = long((phase * length) / (1/30)) * (1/30)   (sample_time_interval, motion_lib_base.py:413-421,
       what HumanoidIm._sample_time uses, humanoid_im.py:652-654).  _reset_ref_state_init (humanoid_im.py:920-926) also clears the
       per-env clock offset, global offset and cycle counter: reset_start_offsets[e] = 0, reset_global_offset[e,:] = 0,
       reset_clear2[e] = 0 when given (and the state is then evaluated with them at zero). */
    int32_t reset_time_interval;
    float* reset_start_offsets; float* reset_global_offset; int64_t* reset_clear2;
} pulse_motion_state_args;
int pulse_sizeof_motion_state_args(void);
int pulse_motion_state(const pulse_motion_state_args* args, pulse_stream_t s);

/* ------------------------------------------------------------------------- *
 * 2c. Per-step rollout bookkeeping of play_steps (phc/learning/amp_agent.py:372-412,
 *     common_agent.py:318-347) in ONE launch instead of ~45 elementwise / reduction launches:
 *       rewards[n]     = (reward + shift) * scale                 (DefaultRewardsShaper)
 *       dones[n]       = dones
 *       next_values[n] = unnorm(critic(next_obs)) * (1 - terminate)          (:394-398)
 *       current_rewards += reward; current_lengths += 1
 *       game_rewards.update(current_rewards[done]); game_lengths.update(current_lengths[done])
 *                       (rl_games AverageMeter: windowed running mean, Appendix B)
 *       current_* *= (1 - done);  done_mask = dones != 0
 *     Buffer element of env e lives at base[e * env_stride] (the experience buffer is env-major).
 * ------------------------------------------------------------------------- */
typedef struct pulse_rollout_record_args {
    int32_t num_envs;
    const float* rewards;             /* (N) env reward buffer */
    float reward_scale, reward_shift;
    const int64_t* dones;             /* (N) reset_buf */
    const int64_t* terminate;         /* (N) extras['terminate'] */
    const float* value_raw; int64_t value_stride;       /* critic output for the NEXT obs, (N, stride) */
    const double* value_mean; const double* value_var;  /* value normaliser statistics (NULL: no un-normalisation) */
    float value_eps;
    float* buf_rewards; float* buf_next_values; uint8_t* buf_dones; int64_t env_stride;   /* slot n of the experience buffer */
    float* current_rewards; float* current_lengths;      /* (N) episode accumulators */
    float* meter_rewards; float* meter_lengths;          /* each: [mean, current_size] (AverageMeter state) */
    float meter_max_size;                                /* games_to_track */
    uint8_t* done_mask;               /* (N) out: dones != 0 (drives the next masked reset) */
    uint8_t* buf_terminate;           /* optional, slot n like buf_dones: terminate != 0, for a bootstrap pass done after the rollout */
    /* optional: meter_blocks workgroups share the envs and leave (finished episodes, their return sum, their length sum, 0) per workgroup in
       meter_partials[block * 4 ..] INSTEAD of updating the meters; pulse_rollout_meters applies the deferred updates of a whole rollout in
       step order (AverageMeter is only read between epochs) */
    float* meter_partials; int32_t meter_blocks;
} pulse_rollout_record_args;
int pulse_sizeof_rollout_record_args(void);
int pulse_rollout_record(const pulse_rollout_record_args* args, pulse_stream_t s);
/* partials: (steps, blocks, 4) as left by ``steps`` pulse_rollout_record launches: the two AverageMeter updates of every step, in order */
int pulse_rollout_meters(const float* partials, int32_t steps, int32_t blocks, float* meter_rewards, float* meter_lengths, float meter_max_size,
                         pulse_stream_t s);

/* Physics STAND-IN (Isaac Gym is out of scope): the simulated humanoid tracks a reference state with a recorded
 * perturbation -- rb = target + noise (quaternions re-normalised), dof_pos/vel = target + noise, dof_force = recorded.
 * Bench / test infrastructure for the motion-library path, kept in the library so the stand-in costs one launch. */
int pulse_kinematic_sim_step(const float* target_rb, const float* noise_rb, float* rb, int64_t num_envs, int32_t num_bodies,
                             const float* target_dof_pos, const float* noise_dof_pos, float* dof_pos,
                             const float* target_dof_vel, const float* noise_dof_vel, float* dof_vel,
                             const float* force_src, float* dof_force, int32_t num_dof, pulse_stream_t s);

/* Physics STAND-IN whose state DEPENDS ON THE ACTION (return-parity experiments; Isaac Gym is out of scope): every joint j carries a
 * 3-d error state e_j (simulated minus reference exp-map dof) driven by PD control towards  sag_j + action_scale * action_j  with a
 * recorded disturbance, integrated semi-implicitly over ``substeps`` per control step:
 *     acc = kp (sag + action_scale a - e) - kd e' + noise;   e' += h acc;   e += h e'            (h = dt / substeps)
 * Body b >= 1 (joint b - 1) is the tracked reference body displaced by that error:
 *     rot = exp_map_to_quat(e) * ref_rot,  pos = ref_pos + lever (e x u_b),  vel = ref_vel + lever (e' x u_b),  ang = ref_ang + e'
 * the root follows the reference; dof_pos / dof_vel = reference + e / e'; dof_force = the PD torque.  reset_mask (optional, bytes):
 * masked envs restart with e = e' = 0.  One thread per (env, body).  The CPU twin is oracle/pd_sim_oracle.py. */
typedef struct pulse_pd_sim_args {
    int64_t num_envs; int32_t num_bodies;            /* J; num_dof = 3 (J - 1) */
    const float* target_rb;                          /* (num_envs, J, 13) reference at the next control step */
    const float* target_dof_pos; const float* target_dof_vel;   /* (num_envs, 3 (J-1)) */
    const float* action;                             /* (num_envs, 3 (J-1)) pd targets (clamped actions) */
    const float* noise_acc;                          /* (num_envs, 3 (J-1)) recorded disturbance */
    const float* sag;                                /* (3 (J-1)) */
    const float* lever_dir;                          /* (J, 3) unit vectors u_b */
    float kp, kd, dt, action_scale, lever; int32_t substeps;
    float* err; float* err_vel;                      /* (num_envs, 3 (J-1)) state, updated in place */
    float* rb; float* dof_pos; float* dof_vel; float* dof_force;   /* outputs */
    const uint8_t* reset_mask;
} pulse_pd_sim_args;
int pulse_sizeof_pd_sim_args(void);
int pulse_pd_sim_step(const pulse_pd_sim_args* args, pulse_stream_t s);

/* ------------------------------------------------------------------------- *
 * 2a''. Trajectory following over a height field (HumanoidTraj / HumanoidPedestrianTerrain, the README's terrain-traversal command):
 *     task observation = ten trajectory samples in the heading frame (compute_location_observations,
 *     humanoid_pedestrian_terrain.py:1587-1616) + the height map under a 32 x 32 sensor grid (get_heights / get_center_heights /
 *     Terrain.sample_height_points, :690-772, 1191-1270; clip and scale of _compute_task_obs :414-427), location (+ power) reward
 *     (:871-890, 1619-1646), and both variants of compute_humanoid_reset (:1476-1531, humanoid_traj.py:256-300).  One 256-thread
 *     workgroup per env.  ``what`` uses the PULSE_TASK_* bits.
 * ------------------------------------------------------------------------- */
typedef struct pulse_traj_step_args {
    uint32_t what; int32_t num_envs;
    const int64_t* env_ids; int32_t num_ids; const uint8_t* env_mask;
    const float* rb; int64_t rb_env_stride; int32_t num_bodies;
    int32_t upright_start;                       /* 0: remove_base_rot before every heading */
    const int64_t* progress; float dt;           /* time = progress * dt */
    /* TrajGenerator state: verts (num_envs, num_verts, 3); traj_dur = num_verts * (episode_dur / (num_verts - 1)) */
    const float* verts; int32_t num_verts; float traj_dur;
    int32_t num_samples; float sample_timestep;  /* numTrajSamples (10), trajSampleTimestep (0.5) */
    /* height sensor: grid points (num_height_points, 2) in the sensor body's heading frame; sensor_body = Head (terrain_obs_root) or 0.
       heightsamples NULL = terrainType 'plane' (zero heights); num_height_points 0 = no terrain observation (HumanoidTraj). */
    const int16_t* heightsamples; int32_t map_rows, map_cols; float horizontal_scale, vertical_scale;
    const float* height_points; int32_t num_height_points; int32_t sensor_body;
    const float* center_points; int32_t num_center_points; int32_t use_center_height; float height_meas_scale;   /* centre grid: 1 .. 256 points */
    /* reward */
    const float* dof_force; const float* dof_vel; int32_t num_dof; float power_coef; int32_t power_reward; int32_t fuzzy_target;
    /* reset */
    const float* contact_forces; const int32_t* contact_body_ids; int32_t num_contact_ids; const float* termination_heights;
    float max_episode_length, fail_dist; int32_t enable_early_termination, terrain_reset, disable_collision;
    /* outputs */
    float* obs; int64_t obs_stride; int32_t obs_offset;
    float* rew; float* rew_raw;                  /* rew_raw (num_envs, 2) = [location reward, power reward] or NULL */
    int64_t* reset; int64_t* terminate;
} pulse_traj_step_args;
int pulse_sizeof_traj_step_args(void);
int pulse_traj_step(const pulse_traj_step_args* args, pulse_stream_t s);

/* TrajGenerator.reset (phc/utils/traj_generator.py:60-123) for the masked envs; the uniform draws are supplied in the reference's order.
   dtheta_scale = dtheta_max * seg_dt, dspeed_scale = accel_max * seg_dt, seg_dt = episode_dur / (num_verts - 1). */
typedef struct pulse_traj_gen_args {
    int32_t num_envs, num_verts; const uint8_t* env_mask;
    const float* rb; int64_t rb_env_stride;      /* trajectory starts at the root's xy */
    const float* u_dtheta; const float* u_sharp; const uint8_t* sharp_mask; const float* u_heading; const float* u_dspeed; const float* u_speed0;
    float dtheta_scale, dspeed_scale, seg_dt, speed_min, speed_max;
    float* verts;
} pulse_traj_gen_args;
int pulse_traj_generate(const pulse_traj_gen_args* args, pulse_stream_t s);

/* ------------------------------------------------------------------------- *
 * 3. GAE: CommonAgent.discount_values + returns, phc/learning/common_agent.py:493-505,
 *    :346-347.  Element (t, n) of every array lives at t*stride_t + n*stride_n.
 *    gamma_tau is the host-side double product gamma*tau rounded to float (as
 *    Python evaluates `self.gamma * self.tau * not_done`).
 * ------------------------------------------------------------------------- */
int pulse_gae(const float* rewards, const float* values, const float* next_values, const uint8_t* dones,
              int32_t horizon, int32_t num_envs, int64_t stride_t, int64_t stride_n,
              float gamma, float gamma_tau, float* advs, float* returns, pulse_stream_t s);

/* ------------------------------------------------------------------------- *
 * 4. FP32 MFMA GEMM with fused epilogues: every nn.Linear forward / backward of
 *    the actor, critic and VAE MLPs (phc/learning/network_builder.py:105-124,245-261;
 *    amp_network_builder.py:127-148,206-211; amp_network_z_builder.py:469-580).
 *    C[m][n] = sum_k A(m,k) * B(n,k), v_mfma_f32_32x32x2_f32 (exact fp32).
 * ------------------------------------------------------------------------- */
#define PULSE_GEMM_RED_CONTIG 0 /* operand stored [out][k]: reduction index contiguous */
#define PULSE_GEMM_OUT_CONTIG 1 /* operand stored [k][out]: output index contiguous   */
#define PULSE_ACT_NONE 0
#define PULSE_ACT_RELU 1
#define PULSE_ACT_SILU 2
#define PULSE_ACT_SILU_D 3    /* SiLU whose C2 (required) receives d silu / d z = s (1 + z (1 - s)), s = sigmoid(z), instead of z: the backward
                                 pass multiplies by it (PULSE_EPI_MUL_AUX) -- the same value the SILU_GRAD epilogue recomputes from z, without the
                                 exp and the division in the launch that has no other work to hide them under (pulse_gemm_f32 only) */
#define PULSE_EPI_BIAS_ACT  0 /* C = act(acc + bias[n]); silu may also store the pre-activation in C2 */
#define PULSE_EPI_RELU_GRAD 1 /* C = acc * (aux > 0)           (aux = forward activation)  */
#define PULSE_EPI_SILU_GRAD 2 /* C = acc * silu'(aux)          (aux = forward pre-activation) */
#define PULSE_EPI_MUL_AUX   3 /* C = acc * aux                 (aux = the derivative a PULSE_ACT_SILU_D forward stored; pulse_gemm_f32 only) */

#define PULSE_GEMM_COMPUTE_F32  0
#define PULSE_GEMM_COMPUTE_BF16 1
#define PULSE_GEMM_COMPUTE_F32X3 2

typedef struct pulse_gemm_desc {
    const float* A; const float* B; float* C;
    float* C2;            /* optional second output (pre-activation), EPI_BIAS_ACT + SILU */
    const float* bias;    /* optional (N) */
    const float* aux;     /* gradient epilogues: (M, N) with pitch ldaux */
    int32_t M, N, K;      /* output M x N, reduction K */
    int32_t lda, ldb, ldc, ldc2, ldaux; /* pitches in floats; lda/ldb multiples of 4 */
    int32_t a_layout, b_layout;         /* PULSE_GEMM_*_CONTIG; (OUT, RED) is unsupported */
    int32_t batch;        /* independent problems; operand z uses pointer + z * stride_* */
    int64_t stride_a, stride_b, stride_c, stride_c2, stride_bias, stride_aux;
    int32_t split_k;      /* >1: slab s of the reduction goes to C + s*split_stride (no epilogue) */
    int64_t split_stride;
    int32_t activation;   /* PULSE_ACT_* */
    int32_t epilogue;     /* PULSE_EPI_* */
    /* optional, (OUT, OUT) layouts only (the dW pass, A = dY stored [batch row][out feature]):
       rowsum[z * stride_rowsum + s * split_stride + m] = sum over slab s of A(k, m) -- the BIAS gradient, taken from the
       A fragments the kernel already holds, so no separate column-sum pass over dY is needed. */
    float* rowsum; int64_t stride_rowsum;
    /* PULSE_GEMM_COMPUTE_BF16: operands (fp32 in memory) are rounded to bf16 on the way into LDS and multiplied on
       v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the arithmetic of a bf16 autocast Linear over fp32 master weights
       (phc/learning/amp_agent.py:671); with round_output_bf16 the outputs are rounded to bf16-representable values as well. */
    /* PULSE_GEMM_COMPUTE_F32X3: an fp32 GEMM computed on the bf16 matrix pipe -- every fp32 operand is split EXACTLY into three
       bf16 numbers (8 + 8 + 8 significand bits) on the way into LDS and the six products that matter at fp32 precision are
       accumulated in fp32.  fp32-grade results (same tolerances as PULSE_GEMM_COMPUTE_F32, same exactness / linearity
       properties) at up to 2.67x the fp32 MFMA rate of gfx950.  Inputs must be finite. */
    int32_t compute_type; int32_t round_output_bf16;
    /* optional (v25): the ReLU derivative as a BIT mask instead of a second pass over the activation matrix.  A forward launch
       (PULSE_EPI_BIAS_ACT + PULSE_ACT_RELU) with relu_mask != NULL also records which outputs are positive; a PULSE_EPI_RELU_GRAD launch with
       aux == NULL and relu_mask != NULL takes C = acc * (bit set) -- the same values as with aux = the forward's output (h > 0 <=> z > 0), at
       1/32 of the aux traffic (nn.ReLU's backward, network_builder.py:105-124 / amp_network_builder.py:230-249 through autograd).
       Layout (independent of the tiling that writes or reads it): 32-bit word [((r >> 6) * 8 + (r & 7)) * ld_mask + (c >> 2)] of batch z's
       mask (relu_mask + z * stride_mask words), bit 4 * ((r >> 3) & 7) + (c & 3) <-> output (r, c).  The buffer holds
       roundup64(M) / 8 * ld_mask words per batch-strided matrix, ld_mask >= roundup4(N) / 4; needs 16-byte aligned C / pitches. */
    uint32_t* relu_mask; int32_t ld_mask; int64_t stride_mask;
} pulse_gemm_desc;

int pulse_sizeof_gemm_desc(void);
int pulse_gemm_f32(const pulse_gemm_desc* desc, pulse_stream_t s);
/* Diagnostics used by tools/gemm_bench and bench.py's clock probe (no effect on results).  THREAD-LOCAL state (like the error string): they
 * affect the pulse_gemm_f32 launches issued by the calling host thread only; never set by the product path (pulse_amd/ calls them only
 * from bench.py's --clock-probe and tools/).  Option 1 = extra dynamic-LDS bytes per workgroup, option 2 = 1 disables the 64-row tile (occupancy
 * experiments), option 3 = tile of the bf16-storage launches of pulse_gemm_x3p (0 automatic, 1 never 256 x 256, 2 256 x 256 whenever N > 128;
 * same results either way up to accumulation order), option 4 = tile of the fp32 (x3) launches of pulse_gemm_f32 (0 = the launcher's cost model,
 * also steered by PULSE_X3_WIDE = 0 / 1 / 2 in the environment; 1 = 128 x 128 only; 2 = 256 x 256 whenever M, N > 128; BIT-IDENTICAL matrix outputs
 * either way, the weight-gradient form's row sums agree to rounding), option 5 = 1: never split a narrow column tail off a 256 x 256 launch (A/B;
 * same bits), option 6 = 1: never the skinny-N kernel (N <= 96 over a long M; same bits), option 7: timing variants of the skinny-N kernel (tools/; results are garbage),
 * option 8 = 1: pulse_ppo_loss in its plain per-sample form instead of the hoisted one (same bits), option 9 = 1: every epilogue row of pulse_gemm_x3p
 * through the general form instead of the round-once rows of the ReLU forward / ReLU-gradient launches (same bits; keys 0 .. 15 are accepted).  With a debug buffer set (8 int64 per workgroup, device memory) every workgroup of the following launches stamps
 * s_memtime / the 100 MHz wall clock at its start, after the main loop and at its end, plus its HW_ID / XCC_ID.  The 256 x 256 bf16-storage
 * kernel of pulse_gemm_x3p honours the same buffer with wall-clock stamps only: [0] start, [1] first stage landed, [2] main loop done,
 * [3] epilogue stores issued, [4] stores acknowledged, [5] XCC_ID, [6] / [7] transpose image written / stores issued of the last epilogue
 * half (tools/gemm_b16_phases.py).  The 256 x 256 x3 kernel: [0] / [1] start (cycles / wall), [2] / [3] main loop done, [4] / [5] epilogue's
 * stores issued, [7] stores acknowledged (wall; tools/gemm_x3w_phases.py). */
int pulse_gemm_set_option(int key, int value);
int pulse_gemm_set_debug_buffer(long long* device_buffer);
/* Tile rows (64 / 128 / 256; 96 = the skinny-N kernel gemm_x3s_kernel) of the calling thread's last pulse_gemm_f32 launch: which kernel served it (256 = gemm_x3w_kernel).  Diagnostics
 * (bench.py attributes its per-launch event times to the kernel that ran); no effect on results. */
int pulse_gemm_last_tile(void);
/* The tiling rule in effect for the calling thread's fp32 (x3) launches: 0 = the launcher's cost model, 1 = 128 x 128 only (option 4 = 1,
 * PULSE_X3_WIDE=0 in the environment, or a device that refused the wide tile's 135 KB of LDS), 2 = 256 x 256 whenever M, N > 128.  Planners that
 * size split-K slab counts for a tiling (pulse_amd/kernels.py: dw_split_x3) ask this instead of re-reading the environment. */
int pulse_gemm_x3_mode(void);
/* ------------------------------------------------------------------------- *
 * 4b. The same fp32-grade GEMM (PULSE_GEMM_COMPUTE_F32X3 arithmetic) over operands kept PRE-SPLIT in HBM: a matrix is stored as
 *     three bf16 "planes" (x = p0 + p1 + p2 exactly, p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1), round to nearest
 *     even); element (r, c) of plane p lives at base[p * plane_stride + r * ld + c] (uint16 bit patterns).  Whoever produces a
 *     matrix writes its planes once (pulse_split_planes, the Cp output of this GEMM, the optimiser step for the weights); the GEMM
 *     main loop is LDS-DMA + ds_read + MFMA only.  Same reference call sites as section 4.
 *     Rules: pitches multiples of 8 elements; the k extent of a reduction-contiguous operand is zero-padded to a multiple of 32
 *     inside its pitch (pulse_split_planes and the Cp epilogue write those zeros); rows / outs past M / N are not read.
 * ------------------------------------------------------------------------- */
typedef struct pulse_gemm_x3p_desc {
    const void* A; int64_t a_plane_stride; int32_t lda;   /* planes of A(m, k); bf16 elements */
    const void* B; int64_t b_plane_stride; int32_t ldb;   /* planes of B(n, k) */
    int32_t a_layout, b_layout;                           /* PULSE_GEMM_*_CONTIG: (red, red), (red, out) or (out, out) as in section 4 */
    float* C; int32_t ldc;                                /* optional fp32 output */
    void* Cp; int64_t c_plane_stride; int32_t ldcp;       /* optional: the output's own planes (columns [N, roundup8(N)) zero-filled) */
    float* C2; int32_t ldc2;                              /* optional pre-activation output (EPI_BIAS_ACT + SILU) */
    const float* bias; const float* aux; int32_t ldaux;
    int32_t M, N, K;
    int32_t batch;
    int64_t stride_a, stride_b, stride_c, stride_cp, stride_c2, stride_bias, stride_aux;   /* per-array element units */
    int32_t split_k; int64_t split_stride;
    int32_t activation, epilogue;                         /* PULSE_ACT_*, PULSE_EPI_* */
    float* rowsum; int64_t stride_rowsum;
    int32_t planes;                                       /* 3 (or 0): fp32-grade, three planes per operand.  1: the operands are plain bf16
                                                             matrices (plane strides ignored), products accumulate in fp32, results leave
                                                             rounded to bf16 (bf16 autocast semantics; split-K slabs stay fp32), Cp is one
                                                             bf16 matrix.  [red][out] operands (a/b_layout OUT_CONTIG) are read through the
                                                             LDS transposing load in either mode. */
    int32_t aux_is_bf16;                                  /* aux points at a bf16 matrix (ldaux / stride_aux in bf16 elements) */
    /* optional: per-row-tile column sums of the OUTPUT as stored (after activation / mask / bf16 rounding):
       out_colsum[z * stride_out_colsum + tile_m * ld_out_colsum + n], tile_m = 0 .. pulse_gemm_x3p_row_tiles(M, N, batch) - 1.  The input-
       gradient launch that produces a layer's dZ hands over the layer's bias gradient (sum the tile rows) without a pass over dZ. */
    float* out_colsum; int64_t stride_out_colsum; int32_t ld_out_colsum;
    /* optional (v28): the ReLU derivative as a bit mask, as in pulse_gemm_desc.relu_mask but in the byte layout the planar epilogue's threads own
       (eight consecutive columns of one row): byte [r * ld_mask8 + (c >> 3)] of batch z's mask (relu_mask8 + z * stride_mask8 bytes), bit (c & 7) <->
       output (r, c).  A PULSE_EPI_BIAS_ACT + PULSE_ACT_RELU launch with relu_mask8 != NULL records it (after the bf16 rounding of single-plane mode:
       exactly the stored activation's sign); a PULSE_EPI_RELU_GRAD launch with aux == NULL reads it instead of the activation matrix -- 1 / 16 of the
       bytes of a bf16 aux, 1 / 32 of an fp32 one; same values.  ld_mask8 >= roundup8(N) / 8. */
    uint8_t* relu_mask8; int32_t ld_mask8; int64_t stride_mask8;
} pulse_gemm_x3p_desc;
/* number of row tiles (256 or 128 rows) a launch of this shape uses: the row count of out_colsum */
int pulse_gemm_x3p_row_tiles(int32_t M, int32_t N, int32_t batch);
int pulse_sizeof_gemm_x3p_desc(void);
int pulse_gemm_x3p(const pulse_gemm_x3p_desc* desc, pulse_stream_t s);
/* fp32 (rows x cols, pitch ld_in floats) -> planes of rows_out x cols_out with pitch ld_out (multiple of 8; columns [cols_out, ld_out)
 * zero-filled).  transpose != 0: out(r, c) = in(c, r).  row_idx (optional, no transpose): out row r = in row row_idx[r].
 * plane_stride 0: write plane 0 only, i.e. the matrix rounded to bf16 (the single-plane mode's operand format). */
int pulse_split_planes(const float* in, int64_t ld_in, int32_t rows_out, int32_t cols_out, void* out, int64_t plane_stride, int32_t ld_out,
                       int32_t transpose, const int64_t* row_idx, pulse_stream_t s);

/* ------------------------------------------------------------------------- *
 * 4d. Glue of the bf16-STORAGE training path (mixed_precision: the reference wraps calc_gradients in autocast, amp_agent.py:671,
 *     common_agent.py:426,461): activations, gradients and per-step weight copies are bf16 matrices feeding pulse_gemm_x3p(planes = 1).
 * ------------------------------------------------------------------------- */
/* out[z][c][r] = bf16(in[z][r][c]) for z < batch: W[out][in] -> W^T[in][out], so an input-gradient GEMM (nn.Linear backward,
 * network_builder.py:105-124) reads both operands reduction-contiguous.  Columns [rows_in, ld_out) of the output rows are not written
 * (allocate them zero: they are the k padding of the consumer). */
int pulse_transpose_to_b16(const float* in, int64_t ld_in, int32_t rows_in, int32_t cols_in, void* out, int64_t ld_out, int32_t batch,
                           int64_t stride_in, int64_t stride_out, pulse_stream_t s);
/* The bf16 images a mixed-precision training pass needs of a flat fp32 parameter buffer, in ONE launch (v20): flat16[i] = bf16(flat[i]) for
 * i < count (elements up to roundup8(count) written as zero; flat16 holds roundup8(count) elements) plus up to four transposed images, each
 * exactly what pulse_transpose_to_b16 writes (`in` usually points into `flat`).  What autocast's per-op weight casts and the .t() views of
 * nn.Linear's backward amount to (common_agent.py:426,461; network_builder.py:105-124), done once per optimiser step's worth of passes. */
typedef struct pulse_b16_transpose {
    const float* in; int64_t ld_in; int32_t rows; int32_t cols;      /* in[z][r][c], r < rows, c < cols */
    void* out; int64_t ld_out;                                        /* out[z][c][r] (bf16) */
    int32_t batch; int32_t reserved; int64_t stride_in; int64_t stride_out;
} pulse_b16_transpose;
int pulse_sizeof_b16_transpose(void);
int pulse_weights_to_b16(const float* flat, int64_t count, void* flat16, int32_t num_transposes, const pulse_b16_transpose* transposes, pulse_stream_t s);
/* pulse_colsum_partial over a bf16 matrix (ld in bf16 elements, multiple of 8): bias-gradient partials, fp32 accumulation.  With
 * num_chunks = the split-K count and partial = the gradient slabs (ld_partial = the slab stride) the bias gradient rides the slab reduce
 * the weight gradients need anyway (what pulse_gemm_desc.rowsum does for the fp32-storage kernels). */
int pulse_colsum_partial_b16(const void* x, int32_t m, int32_t n, int64_t ld, int32_t num_chunks, float* partial, int64_t ld_partial,
                             pulse_stream_t s);
/* partial[c][n] = sum over chunk c's rows of w[m * w_stride] * x[m][n] (w bf16): the weight gradient of a ONE-output Linear -- the
 * discriminator's logit layer (amp_network_builder.py:233-237: d w3 = sum_m dlogit[m] h2[m][:]) -- without a 1 x n GEMM. */
int pulse_colsum_weighted_b16(const void* x, int32_t m, int32_t n, int64_t ld, const void* w, int64_t w_stride, int32_t num_chunks,
                              float* partial, int64_t ld_partial, pulse_stream_t s);
/* AMPAgent._disc_loss gradient penalty (amp_agent.py:925-934) on g = dD/dx of the demo rows (rows x cols fp32, pad columns zero):
 * partials[block] = sum of g^2 over the block's share; out32 / out16 (either may be NULL) = scale * g, the seed of the penalty's backward
 * pass (scale = 2 disc_grad_penalty disc_coef / (world_size rows)). */
int pulse_disc_penalty(const float* g, int64_t ldg, int32_t rows, int32_t cols, float scale, float* out32, int64_t ld32, void* out16,
                       int64_t ld16, float* partials, int32_t num_blocks, pulse_stream_t s);
/* disc_logit_reg / disc_weight_decay terms (amp_agent.py:919-923, 936-940) over up to four ranges of a flat parameter buffer:
 * grad[off_r + i] += alpha_r * flat[off_r + i] (grad may be NULL), partials[block * 4 + r] = the block's share of sum flat[range r]^2.
 * offsets / lengths / alphas are HOST arrays of num_ranges entries. */
int pulse_disc_reg(const float* flat, float* grad, int32_t num_ranges, const int64_t* offsets, const int64_t* lengths, const float* alphas,
                   float* partials, int32_t num_blocks, pulse_stream_t s);
/* The split-K slab reduce of a whole flat gradient buffer in one launch, region by region (each region = a range of the flat layout with the
 * number of slabs its weight-gradient launch wrote): out[off + i] = scale * sum_s slabs[s * slab_stride + off + i] + alpha_r * flat[off + i]
 * (the regularisers of pulse_disc_reg; alphas / flat may be NULL).  sq_partials[block] (optional) = the block's share of sum out^2 -- the
 * clip_grad_norm_ input (common_agent.py:472-478) without a pass of its own -- and w2_partials[block * 8 + r] (optional) the share of
 * sum flat[region r]^2 for the first 8 regions.  offsets / counts / nslabs / alphas are HOST arrays of num_regions <= 32 entries (8 before v29); offsets
 * and counts multiples of 4.  nslabs[r] = 0 (v29): no launch of this pass wrote the region -- out is alpha_r * flat there (zero without a
 * regulariser), nothing is read from the slabs: a pass that visits part of a network (the kin pass of PULSE: encoder / decoder / prior, not the
 * critic) neither zero-fills nor sums the slabs of the rest.
 * region_src / region_src_stride (HOST arrays, may be NULL; v19): a region with region_src[r] != NULL sums nslabs[r] partial rows of its OWN device
 * buffer (row s at region_src[r] + s * region_src_stride[r]) instead of the slabs -- column-sum partials of a bias gradient, or the scratch of a
 * weight gradient that was split wider than the slab count; same summation order as pulse_reduce_slabs (bit-identical results). */
int pulse_reduce_grads(const float* slabs, int64_t slab_stride, int32_t num_regions, const int64_t* offsets, const int64_t* counts,
                       const int32_t* nslabs, const float* alphas, const float* const* region_src, const int64_t* region_src_stride,
                       float* out, float scale, const float* flat, float* sq_partials, float* w2_partials, int32_t num_blocks, pulse_stream_t s);
/* AMPAgent._calc_disc_rewards (amp_agent.py:1027-1041): out[i * out_stride] = -log(max(1 - sigmoid(logits[i * logit_stride]), 1e-4)) * scale */
int pulse_disc_reward(const float* logits, int64_t logit_stride, int64_t n, float scale, float* out, int64_t out_stride, pulse_stream_t s);

/* ------------------------------------------------------------------------- *
 * 4c. PULSE VAE head algebra (no autograd on the product path): form_embedding (amp_network_z_builder.py:79-121), the losses of
 *     AMPAgent._optimize_kin (amp_agent.py:771-849; kl_multi loss_functions.py:3-10) and their head-level gradients.
 *     Heads rows are [mu (E) | raw logvar (E)]; clamp_logvar applies clamp(logvar, -5, clamp_max) (use_vae_clamped_prior).
 * ------------------------------------------------------------------------- */
typedef struct pulse_vae_embed_args {
    const float* heads; int64_t heads_stride;   /* encoder heads (rows, >= 2E) */
    const float* eps; int64_t eps_stride;       /* re-parameterisation noise (rows, E) or NULL (z = mu, flags.test) */
    const float* x; int64_t x_stride;           /* normalised observation; its first self_obs_size columns are copied */
    float* ain; int64_t ain_stride;             /* decoder input: [self obs | .. | z at column z_col] */
    float* cin; int64_t cin_stride;             /* optional: critic input, self obs columns only */
    int32_t rows, embedding_size, self_obs_size, z_col;
    int32_t clamp_logvar; float clamp_max;
} pulse_vae_embed_args;
int pulse_vae_embed(const pulse_vae_embed_args* args, pulse_stream_t s);

typedef struct pulse_vae_kin_args {
    const float* pred; int64_t pred_stride;     /* decoder output mu (rows, A) */
    const float* gt; int64_t gt_stride;         /* kin_dict['gt_action'] */
    const float* zheads; int64_t zheads_stride;
    const float* pheads; int64_t pheads_stride; /* learned prior heads */
    const int64_t* progress;                    /* kin_dict['progress_buf'] (rows), env-major sequences of ``horizon`` steps */
    int32_t rows, num_actions, embedding_size, horizon;
    int32_t clamp_logvar; float clamp_max;
    int32_t use_ar1, use_regu;
    float* dmu; int64_t dmu_stride;             /* out: d mean||pred - gt|| / d pred */
    float* partials; int32_t num_blocks;        /* out: (num_blocks, 8) sums: action norm, KL, AR(1) norm, prior_mu^2, vae_mu^2, prior_lv^2, vae_lv^2, 0 */
} pulse_vae_kin_args;
int pulse_vae_kin_loss(const pulse_vae_kin_args* args, pulse_stream_t s);

typedef struct pulse_vae_head_bwd_args {
    const float* zheads; int64_t zheads_stride;
    const float* pheads; int64_t pheads_stride; /* NULL: no KL / regulariser terms (the PPO backward through the VAE) */
    const float* eps; int64_t eps_stride;       /* the noise the forward used (NULL: z = mu) */
    const float* dz; int64_t dz_stride;         /* d loss / d z from the decoder's input gradient (NULL: none) */
    const int64_t* progress;
    int32_t rows, embedding_size, horizon;
    int32_t clamp_logvar; float clamp_max;
    float c_kl;    /* kld_coefficient / rows */
    float c_ar1;   /* ar1_coefficient / (sequences * (horizon - 1)) or 0 */
    float c_regu;  /* 0.005 * 0.001 / (rows * E) or 0 */
    float* dzheads; int64_t dzheads_stride;     /* out (rows, 2E) */
    float* dpheads; int64_t dpheads_stride;     /* out (rows, 2E), optional */
} pulse_vae_head_bwd_args;
int pulse_vae_head_backward(const pulse_vae_head_bwd_args* args, pulse_stream_t s);
int pulse_sizeof_vae_embed_args(void);
int pulse_sizeof_vae_kin_args(void);
int pulse_sizeof_vae_head_bwd_args(void);

/* out[i] = scale * sum_s slabs[s*slab_stride + i]  (deterministic split-K / partial-sum reduction) */
int pulse_reduce_slabs(const float* slabs, int32_t num_slabs, int64_t slab_stride, int64_t count, float* out,
                       float scale, pulse_stream_t s);
/* partial[c*ld_partial + n] = sum of rows of chunk c of x (m x n, pitch ld): bias-gradient partials */
int pulse_colsum_partial(const float* x, int32_t m, int32_t n, int32_t ld, int32_t num_chunks, float* partial,
                         int64_t ld_partial, pulse_stream_t s);

/* ------------------------------------------------------------------------- *
 * 5. Learner-side elementwise / reduction kernels (PPO update).
 * ------------------------------------------------------------------------- */
/* RunningMeanStd.forward, phc/utils/running_mean_std.py:69-109.
 *   mode 0: y = clamp((x - mean) / sqrt(var + eps), -clip, clip);  mode 1 (unnorm): y = sqrt(var+eps)*clamp(x)+mean.
 *   x rows may be gathered: row i of y comes from x row row_idx[i] (NULL = identity) -- this is the
 *   AMPDataset minibatch gather (phc/learning/amp_datasets.py:81-94) fused with the normaliser.
 *   Columns [cols, y_cols) of y are zero-filled (GEMM-ready pitch).  If moment_partials != NULL the
 *   per-column batch sums / sums of squares (fp64) of the RAW rows are written to
 *   moment_partials[block][2][cols] for pulse_rms_update (the "update after normalise" of :98-107).
 *   mean / var are the module's fp64 buffers. */
int pulse_rms_normalize(const float* x, int64_t x_stride, const int64_t* row_idx, int32_t rows, int32_t cols,
                        const double* mean, const double* var, float eps, float clip, int32_t mode,
                        float* y, int64_t y_stride, int32_t y_cols,
                        double* moment_partials, int32_t num_blocks, pulse_stream_t s);
/* The same normalise pass (mode 0, wide-row form: 64 <= cols, 16-byte aligned rows) also writing the THREE bf16 planes of y (section 4b's
 * operand format; planes[p * plane_stride + row * planes_ld + col], y_cols a multiple of 32, columns [cols, y_cols) zero) -- the
 * layer-1 input of pulse_gemm_x3p split by its producer (same call site: RunningMeanStd.forward in _preproc_obs, common_agent.py:257-261). */
int pulse_rms_normalize_planes(const float* x, int64_t x_stride, const int64_t* row_idx, int32_t rows, int32_t cols,
                               const double* mean, const double* var, float eps, float clip,
                               float* y, int64_t y_stride, int32_t y_cols, double* moment_partials, int32_t num_blocks,
                               void* planes, int64_t plane_stride, int64_t planes_ld, pulse_stream_t s);
/* The same normalise pass (mode 0, wide-row form) whose output IS a bf16 matrix (y16[row * y_stride + col], strides in bf16 elements, columns
 * [cols, y_cols) zero): the layer-1 operand of the bf16-storage training path (section 4d; a bf16 autocast Linear rounds its fp32 input
 * the same way, amp_agent.py:671). */
int pulse_rms_normalize_b16(const float* x, int64_t x_stride, const int64_t* row_idx, int32_t rows, int32_t cols,
                            const double* mean, const double* var, float eps, float clip,
                            void* y16, int64_t y_stride, int32_t y_cols, double* moment_partials, int32_t num_blocks, pulse_stream_t s);
/* pulse_rms_normalize (mode 0, wide-row form) that ALSO stores the raw rows it read: raw_out[i * raw_stride + c] = x[row(i)][c] for c < cols
 * rounded up to a multiple of 4 (16-byte stores: the source's pad columns travel with the last group) -- the rollout's
 * experience_buffer.update_data('obses', n, obs) (a2c_common.play_steps) done by the pass that reads the observation anyway.  v29. */
int pulse_rms_normalize_copy(const float* x, int64_t x_stride, const int64_t* row_idx, int32_t rows, int32_t cols,
                             const double* mean, const double* var, float eps, float clip,
                             float* y, int64_t y_stride, int32_t y_cols, double* moment_partials, int32_t num_blocks,
                             float* raw_out, int64_t raw_stride, pulse_stream_t s);
/* _update_mean_var_count_from_moments, running_mean_std.py:56-67 (unbiased batch variance).
 * count_old is tracked by the host (it only ever grows by the batch size). */
int pulse_rms_update(double* mean, double* var, double* count_out, const double* moment_partials, int32_t num_blocks,
                     int32_t cols, double count_old, double batch_count, pulse_stream_t s);

/* rl_games ModelA2CContinuousLogStd eval branch (3P) + value un-normalisation
 * (phc/learning/common_agent.py:262-288): action = mu + exp(logstd)*noise, neglogp, sigma rows,
 * value = sqrt(var+eps)*clamp(value_raw,+-5)+mean.  Row pitches are explicit so outputs land directly in
 * the experience buffer slot. */
int pulse_policy_sample(const float* mu, int64_t mu_stride, const float* logstd, const float* noise, int64_t noise_stride,
                        const float* value_raw, int64_t value_stride, const double* value_mean, const double* value_var,
                        int32_t rows, int32_t num_actions, float* actions, int64_t actions_stride, float* sigmas,
                        int64_t sigmas_stride, float* neglogp, int64_t neglogp_stride, float* values, int64_t values_out_stride,
                        float* mus_out /* optional copy of mu, e.g. the experience-buffer slot */, int64_t mus_out_stride,
                        pulse_stream_t s);

typedef struct pulse_ppo_loss_args {
    /* network outputs for the minibatch (row i) */
    const float* mu; int64_t mu_stride;
    const float* value; int64_t value_stride;
    const float* logstd;               /* (A) */
    /* dataset tensors, row idx[i] (NULL idx = identity): AMPDataset._get_item gather fused in */
    const int64_t* idx;
    const float* actions; int64_t actions_stride;
    const float* old_mu; int64_t old_mu_stride;
    const float* old_logstd;           /* (A): old sigma is the same state-independent parameter */
    const float* old_neglogp;          /* (B) */
    const float* advantages;           /* (B) */
    const float* old_values;           /* (B) */
    const float* returns;              /* (B) */
    int32_t rows, num_actions;
    float e_clip, critic_coef, bounds_loss_coef; int32_t clip_value; int32_t has_bounds_loss;
    /* outputs */
    float* dmu; int64_t dmu_stride;    /* d loss / d mu      (rows, A) */
    float* dvalue; int64_t dvalue_stride; /* d loss / d value (rows) */
    float* partials;                   /* (num_blocks, 8): sums of a_loss, c_loss, b_loss, clipped, kl, 0,0,0 */
    int32_t num_blocks;
    /* optional bf16 copies of the two gradients (strides in bf16 elements): the operands of the bf16-storage backward GEMMs (section 4d) */
    void* dmu16; int64_t dmu16_stride;
    void* dvalue16; int64_t dvalue16_stride;
} pulse_ppo_loss_args;
/* CommonAgent.calc_gradients loss section + analytic gradients w.r.t. the network outputs:
 * _actor_loss / _critic_loss / bound_loss (phc/learning/common_agent.py:512-520,564-587),
 * neglogp + policy_kl (rl_games 3P).  loss = a + critic_coef*c + bounds_coef*b (entropy term is constant). */
int pulse_sizeof_ppo_loss_args(void);
int pulse_ppo_loss(const pulse_ppo_loss_args* args, pulse_stream_t s);

/* _calc_advs, common_agent.py:589-599: adv = returns - values; (adv - mean)/(std_unbiased + 1e-8). Two launches. */
int pulse_advantage_moments(const float* returns, const float* values, int64_t count, float* adv, double* partials,
                            int32_t num_blocks, pulse_stream_t s);
int pulse_advantage_normalize(float* adv, int64_t count, const double* partials, int32_t num_blocks, pulse_stream_t s);

/* nn.utils.clip_grad_norm_ + torch.optim.Adam.step fused over a flat parameter buffer
 * (common_agent.py:472-478; Adam(eps=1e-8) :66).  sqnorm_partials: per-block sums of squares of the
 * gradient (from pulse_sqnorm_partial); clip coefficient = min(1, max_norm / (norm + 1e-6)); max_norm <= 0
 * disables clipping.  step is the 1-based Adam step count. */
int pulse_sqnorm_partial(const float* x, int64_t count, float* partials, int32_t num_blocks, pulse_stream_t s);

/* AMPAgent._disc_loss head, phc/learning/amp_agent.py:895-952 (disc_loss_neg / disc_loss_pos :954-969, _compute_disc_acc :971-977), on
 * the logits of the 3b stacked rows [agent | replay | demo] (row i at logits[i * logit_stride]):
 *   stats[0] = 0.5 (BCEWithLogits(agent U replay, 0) + BCEWithLogits(demo, 1)),  [1], [2] the two means,
 *   stats[3] = mean(agent logit < 0), [4] = mean(demo logit > 0), [5], [6] = mean agent / demo logit, [7] = 0
 *   dlogits[i * dlogit_stride] = scale * d stats[0] / d logit_i          (scale = disc_coef / world_size) */
int pulse_disc_head(const float* logits, int64_t logit_stride, int32_t b, float scale, float* dlogits, int64_t dlogit_stride,
                    float* stats, pulse_stream_t s);
/* pulse_disc_head also (or only) writing the logit gradients as bf16 (dlogits16[i * dlogit16_stride]): operand of the bf16-storage
 * discriminator backward.  dlogits may be NULL.  bias_grad (optional, v21): *bias_grad = sum_i of the logit gradients as stored (the
 * bf16-rounded ones when dlogits16 is given) = the gradient of the logit layer's bias, so the backward pass needs no column-sum launch for it. */
int pulse_disc_head_b16(const float* logits, int64_t logit_stride, int32_t b, float scale, float* dlogits, int64_t dlogit_stride,
                        void* dlogits16, int64_t dlogit16_stride, float* stats, float* bias_grad, pulse_stream_t s);
int pulse_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t count, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                    const float* sqnorm_partials, int32_t num_partials, float* grad_norm_out, pulse_stream_t s);
/* The same step over up to four flat buffers in ONE launch (v21): one optimiser over several parameter groups with the joint gradient-norm
 * clip (AMPAgent: policy + discriminator).  params / grads / exp_avg / exp_avg_sq / counts are HOST arrays of num_groups entries; per element
 * exactly pulse_adam_step's arithmetic (bit-identical results); grad_norm_out is written once. */
int pulse_adam_step_multi(int32_t num_groups, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                          const int64_t* counts, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                          const float* sqnorm_partials, int32_t num_partials, float* grad_norm_out, pulse_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* PULSE_HIP_H */
