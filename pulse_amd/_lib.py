"""ctypes binding of libpulse_hip.so (the C ABI declared in include/pulse_hip.h).

The library is the product: there is NO CPU / PyTorch fallback.  If the shared
object is missing or cannot be loaded every op raises ``PulseLibraryError``
loudly (build it with ``python pulse_amd/csrc/build.py`` or
``__graft_entry__.build()``).
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# PULSE_HIP_LIB: another build of the SAME library (tools/im_step_repro.py compares compile variants); default = the in-tree build
LIB_PATH = os.environ.get("PULSE_HIP_LIB") or os.path.join(_HERE, "csrc", "libpulse_hip.so")
ABI_VERSION = 29

PULSE_IM_SELF_OBS = 1
PULSE_IM_TASK_OBS = 2
PULSE_IM_REWARD = 4
PULSE_IM_RESET = 8


class PulseLibraryError(RuntimeError):
    pass


class RewardSpecs(Structure):
    _fields_ = [("k_pos", c_float), ("k_rot", c_float), ("k_vel", c_float), ("k_ang_vel", c_float),
                ("w_pos", c_float), ("w_rot", c_float), ("w_vel", c_float), ("w_ang_vel", c_float),
                ("power_coef", c_float), ("power_reward", c_int32)]


class MotionTables(Structure):
    _fields_ = [("frames", c_void_p), ("frame_stride", c_int64), ("total_frames", c_int64), ("num_bodies", c_int32),
                ("off_gts", c_int32), ("off_grs", c_int32), ("off_lrs", c_int32), ("off_gvs", c_int32), ("off_gavs", c_int32),
                ("off_dvs", c_int32),
                ("motion_lengths", c_void_p), ("motion_dt", c_void_p), ("motion_num_frames", c_void_p), ("length_starts", c_void_p),
                ("num_motions", c_int32)]


class ImStepArgs(Structure):
    _fields_ = [
        ("rb", c_void_p), ("rb_env_stride", c_int64), ("num_envs", c_int32), ("num_bodies", c_int32),
        ("env_ids", c_void_p), ("num_ids", c_int32), ("env_mask", c_void_p),
        ("ref_now_pos", c_void_p), ("ref_now_rot", c_void_p), ("ref_now_vel", c_void_p), ("ref_now_ang", c_void_p),
        ("ref_next_pos", c_void_p), ("ref_next_rot", c_void_p), ("ref_next_vel", c_void_p), ("ref_next_ang", c_void_p),
        ("time_steps", c_int32),
        ("dof_force", c_void_p), ("dof_vel", c_void_p), ("num_dof", c_int32),
        ("progress", c_void_p), ("pass_time", c_void_p), ("cycle_counter", c_void_p),
        ("track_ids", c_void_p), ("num_track", c_int32), ("reset_ids", c_void_p), ("num_reset", c_int32),
        ("term_dist", c_void_p), ("reset_use_mean", c_int32), ("full_body_reward", c_int32),
        ("what", c_uint32), ("obs_version", c_int32), ("local_root_obs", c_int32), ("root_height_obs", c_int32),
        ("specs", RewardSpecs),
        ("upright_start", c_int32), ("enable_early_termination", c_int32), ("self_obs_version", c_int32), ("hist_steps", c_int32),
        ("force_sensor", c_void_p), ("force_sensor_width", c_int32), ("dof_pos", c_void_p), ("ref_next_dof_pos", c_void_p),
        ("obs", c_void_p), ("obs_stride", c_int64), ("obs_cols", c_int32),
        ("rew", c_void_p), ("rew_raw", c_void_p), ("reset", c_void_p), ("terminate", c_void_p),
        ("progress_rw", c_void_p), ("progress_inc", c_int32), ("clock_dt", c_float),
        ("clock_start_times", c_void_p), ("clock_start_offsets", c_void_p), ("clock_motion_len", c_void_p),
        ("cycle_motion", c_int32), ("max_episode_length", c_int32), ("pass_time_out", c_void_p),
        ("use_motion", c_int32), ("motion", MotionTables), ("motion_ids", c_void_p), ("motion_offset", c_void_p), ("traj_dt", c_float),
        ("track_rb", c_void_p), ("track_rb_stride", c_int64), ("track_dof_pos", c_void_p), ("track_dof_vel", c_void_p),
        ("smpl_params", c_void_p), ("smpl_params_width", c_int32), ("smpl_params_stride", c_int64),
        ("limb_weights", c_void_p), ("limb_weights_width", c_int32), ("limb_weights_stride", c_int64),
        ("recovery_counter", c_void_p),
        ("zero_out_far", c_int32), ("close_distance", c_float), ("far_distance", c_float), ("point_goal", c_void_p),
        ("occl_bits", c_void_p), ("occl_reset", c_int32),
        ("obs_copy", c_void_p), ("obs_copy_stride", c_int64),
    ]


class TaskStepArgs(Structure):
    _fields_ = [
        ("task", c_int32), ("what", c_uint32), ("num_envs", c_int32), ("env_ids", c_void_p), ("num_ids", c_int32), ("env_mask", c_void_p),
        ("rb", c_void_p), ("rb_env_stride", c_int64), ("num_bodies", c_int32), ("prev_root_pos", c_void_p), ("dt", c_float),
        ("tar_speed", c_void_p), ("tar_pos", c_void_p), ("reach_body_id", c_int32), ("tar_states", c_void_p),
        ("tar_contact_forces", c_void_p), ("strike_body_ids", c_void_p), ("num_strike", c_int32),
        ("contact_forces", c_void_p), ("contact_body_ids", c_void_p), ("num_contact_ids", c_int32), ("termination_heights", c_void_p),
        ("progress", c_void_p), ("max_episode_length", c_float), ("enable_early_termination", c_int32),
        ("dof_force", c_void_p), ("dof_vel", c_void_p), ("num_dof", c_int32), ("power_coef", c_float), ("power_reward", c_int32),
        ("obs", c_void_p), ("obs_stride", c_int64), ("obs_offset", c_int32),
        ("rew", c_void_p), ("rew_raw", c_void_p), ("rew_raw_width", c_int32), ("reset", c_void_p), ("terminate", c_void_p),
    ]


TASK_SPEED, TASK_REACH, TASK_STRIKE = 1, 2, 3
TASK_OBS, TASK_REWARD, TASK_RESET = 1, 2, 4


class AmpObsArgs(Structure):
    _fields_ = [("rb", c_void_p), ("rb_env_stride", c_int64), ("dof_pos", c_void_p), ("dof_vel", c_void_p), ("num_dof", c_int32),
                ("num_envs", c_int32), ("env_ids", c_void_p), ("num_ids", c_int32), ("env_mask", c_void_p),
                ("joint_ids", c_void_p), ("num_joints", c_int32), ("zero_joint_mask", c_uint32),
                ("key_body_ids", c_void_p), ("num_key_bodies", c_int32), ("local_root_obs", c_int32), ("root_height_obs", c_int32),
                ("out", c_void_p), ("out_stride", c_int64),
                ("hist_steps", c_int32), ("window_out", c_void_p), ("window_stride", c_int64)]


class B16Transpose(Structure):
    _fields_ = [("in_", c_void_p), ("ld_in", c_int64), ("rows", c_int32), ("cols", c_int32), ("out", c_void_p), ("ld_out", c_int64),
                ("batch", c_int32), ("reserved", c_int32), ("stride_in", c_int64), ("stride_out", c_int64)]


class AmpHistArgs(Structure):
    _fields_ = [("tab", MotionTables), ("motion_ids", c_void_p), ("start_times", c_void_p), ("dt", c_float),
                ("num_envs", c_int32), ("env_mask", c_void_p), ("hist_steps", c_int32),
                ("joint_ids", c_void_p), ("num_joints", c_int32), ("key_body_ids", c_void_p), ("num_key_bodies", c_int32),
                ("local_root_obs", c_int32), ("root_height_obs", c_int32),
                ("hist", c_void_p), ("env_stride", c_int64), ("step_stride", c_int64)]


class MotionStateArgs(Structure):
    _fields_ = [("tab", MotionTables), ("n", c_int64), ("motion_ids", c_void_p), ("motion_times", c_void_p),
                ("progress", c_void_p), ("step_shift", c_int32), ("dt", c_float), ("start_times", c_void_p), ("start_offsets", c_void_p),
                ("time_steps", c_int32), ("traj_dt", c_float), ("offset", c_void_p), ("root_only", c_int32),
                ("rg_pos", c_void_p), ("rb_rot", c_void_p), ("body_vel", c_void_p), ("body_ang_vel", c_void_p),
                ("dof_pos", c_void_p), ("dof_vel", c_void_p), ("root_pos", c_void_p),
                ("rb_records", c_void_p), ("rb_query_stride", c_int64),
                ("frame_idx0", c_void_p), ("frame_idx1", c_void_p), ("blend", c_void_p),
                ("reset_mask", c_void_p), ("reset_phase", c_void_p), ("reset_start_times", c_void_p), ("reset_progress", c_void_p),
                ("reset_clear0", c_void_p), ("reset_clear1", c_void_p), ("reset_time_interval", c_int32),
                ("reset_start_offsets", c_void_p), ("reset_global_offset", c_void_p), ("reset_clear2", c_void_p)]


class RolloutRecordArgs(Structure):
    _fields_ = [("num_envs", c_int32), ("rewards", c_void_p), ("reward_scale", c_float), ("reward_shift", c_float),
                ("dones", c_void_p), ("terminate", c_void_p), ("value_raw", c_void_p), ("value_stride", c_int64),
                ("value_mean", c_void_p), ("value_var", c_void_p), ("value_eps", c_float),
                ("buf_rewards", c_void_p), ("buf_next_values", c_void_p), ("buf_dones", c_void_p), ("env_stride", c_int64),
                ("current_rewards", c_void_p), ("current_lengths", c_void_p), ("meter_rewards", c_void_p), ("meter_lengths", c_void_p),
                ("meter_max_size", c_float), ("done_mask", c_void_p), ("buf_terminate", c_void_p),
                ("meter_partials", c_void_p), ("meter_blocks", c_int32)]


class PdSimArgs(Structure):
    _fields_ = [("num_envs", c_int64), ("num_bodies", c_int32), ("target_rb", c_void_p), ("target_dof_pos", c_void_p), ("target_dof_vel", c_void_p),
                ("action", c_void_p), ("noise_acc", c_void_p), ("sag", c_void_p), ("lever_dir", c_void_p),
                ("kp", c_float), ("kd", c_float), ("dt", c_float), ("action_scale", c_float), ("lever", c_float), ("substeps", c_int32),
                ("err", c_void_p), ("err_vel", c_void_p), ("rb", c_void_p), ("dof_pos", c_void_p), ("dof_vel", c_void_p), ("dof_force", c_void_p),
                ("reset_mask", c_void_p)]


class TrajStepArgs(Structure):
    _fields_ = [("what", c_uint32), ("num_envs", c_int32), ("env_ids", c_void_p), ("num_ids", c_int32), ("env_mask", c_void_p),
                ("rb", c_void_p), ("rb_env_stride", c_int64), ("num_bodies", c_int32), ("upright_start", c_int32),
                ("progress", c_void_p), ("dt", c_float),
                ("verts", c_void_p), ("num_verts", c_int32), ("traj_dur", c_float), ("num_samples", c_int32), ("sample_timestep", c_float),
                ("heightsamples", c_void_p), ("map_rows", c_int32), ("map_cols", c_int32), ("horizontal_scale", c_float), ("vertical_scale", c_float),
                ("height_points", c_void_p), ("num_height_points", c_int32), ("sensor_body", c_int32),
                ("center_points", c_void_p), ("num_center_points", c_int32), ("use_center_height", c_int32), ("height_meas_scale", c_float),
                ("dof_force", c_void_p), ("dof_vel", c_void_p), ("num_dof", c_int32), ("power_coef", c_float), ("power_reward", c_int32),
                ("fuzzy_target", c_int32),
                ("contact_forces", c_void_p), ("contact_body_ids", c_void_p), ("num_contact_ids", c_int32), ("termination_heights", c_void_p),
                ("max_episode_length", c_float), ("fail_dist", c_float), ("enable_early_termination", c_int32), ("terrain_reset", c_int32),
                ("disable_collision", c_int32),
                ("obs", c_void_p), ("obs_stride", c_int64), ("obs_offset", c_int32), ("rew", c_void_p), ("rew_raw", c_void_p),
                ("reset", c_void_p), ("terminate", c_void_p)]


class TrajGenArgs(Structure):
    _fields_ = [("num_envs", c_int32), ("num_verts", c_int32), ("env_mask", c_void_p), ("rb", c_void_p), ("rb_env_stride", c_int64),
                ("u_dtheta", c_void_p), ("u_sharp", c_void_p), ("sharp_mask", c_void_p), ("u_heading", c_void_p), ("u_dspeed", c_void_p),
                ("u_speed0", c_void_p), ("dtheta_scale", c_float), ("dspeed_scale", c_float), ("seg_dt", c_float), ("speed_min", c_float),
                ("speed_max", c_float), ("verts", c_void_p)]


class VaeEmbedArgs(Structure):
    _fields_ = [("heads", c_void_p), ("heads_stride", c_int64), ("eps", c_void_p), ("eps_stride", c_int64), ("x", c_void_p), ("x_stride", c_int64),
                ("ain", c_void_p), ("ain_stride", c_int64), ("cin", c_void_p), ("cin_stride", c_int64),
                ("rows", c_int32), ("embedding_size", c_int32), ("self_obs_size", c_int32), ("z_col", c_int32),
                ("clamp_logvar", c_int32), ("clamp_max", c_float)]


class VaeKinArgs(Structure):
    _fields_ = [("pred", c_void_p), ("pred_stride", c_int64), ("gt", c_void_p), ("gt_stride", c_int64),
                ("zheads", c_void_p), ("zheads_stride", c_int64), ("pheads", c_void_p), ("pheads_stride", c_int64), ("progress", c_void_p),
                ("rows", c_int32), ("num_actions", c_int32), ("embedding_size", c_int32), ("horizon", c_int32),
                ("clamp_logvar", c_int32), ("clamp_max", c_float), ("use_ar1", c_int32), ("use_regu", c_int32),
                ("dmu", c_void_p), ("dmu_stride", c_int64), ("partials", c_void_p), ("num_blocks", c_int32)]


class VaeHeadBwdArgs(Structure):
    _fields_ = [("zheads", c_void_p), ("zheads_stride", c_int64), ("pheads", c_void_p), ("pheads_stride", c_int64),
                ("eps", c_void_p), ("eps_stride", c_int64), ("dz", c_void_p), ("dz_stride", c_int64), ("progress", c_void_p),
                ("rows", c_int32), ("embedding_size", c_int32), ("horizon", c_int32), ("clamp_logvar", c_int32), ("clamp_max", c_float),
                ("c_kl", c_float), ("c_ar1", c_float), ("c_regu", c_float),
                ("dzheads", c_void_p), ("dzheads_stride", c_int64), ("dpheads", c_void_p), ("dpheads_stride", c_int64)]


class GemmX3pDesc(Structure):
    _fields_ = [("A", c_void_p), ("a_plane_stride", c_int64), ("lda", c_int32),
                ("B", c_void_p), ("b_plane_stride", c_int64), ("ldb", c_int32),
                ("a_layout", c_int32), ("b_layout", c_int32),
                ("C", c_void_p), ("ldc", c_int32),
                ("Cp", c_void_p), ("c_plane_stride", c_int64), ("ldcp", c_int32),
                ("C2", c_void_p), ("ldc2", c_int32),
                ("bias", c_void_p), ("aux", c_void_p), ("ldaux", c_int32),
                ("M", c_int32), ("N", c_int32), ("K", c_int32), ("batch", c_int32),
                ("stride_a", c_int64), ("stride_b", c_int64), ("stride_c", c_int64), ("stride_cp", c_int64), ("stride_c2", c_int64),
                ("stride_bias", c_int64), ("stride_aux", c_int64),
                ("split_k", c_int32), ("split_stride", c_int64), ("activation", c_int32), ("epilogue", c_int32),
                ("rowsum", c_void_p), ("stride_rowsum", c_int64), ("planes", c_int32), ("aux_is_bf16", c_int32),
                ("out_colsum", c_void_p), ("stride_out_colsum", c_int64), ("ld_out_colsum", c_int32),
                ("relu_mask8", c_void_p), ("ld_mask8", c_int32), ("stride_mask8", c_int64)]


class GemmDesc(Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("C2", c_void_p), ("bias", c_void_p), ("aux", c_void_p),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("lda", c_int32), ("ldb", c_int32), ("ldc", c_int32), ("ldc2", c_int32), ("ldaux", c_int32),
        ("a_layout", c_int32), ("b_layout", c_int32), ("batch", c_int32),
        ("stride_a", c_int64), ("stride_b", c_int64), ("stride_c", c_int64), ("stride_c2", c_int64),
        ("stride_bias", c_int64), ("stride_aux", c_int64),
        ("split_k", c_int32), ("split_stride", c_int64), ("activation", c_int32), ("epilogue", c_int32),
        ("rowsum", c_void_p), ("stride_rowsum", c_int64),
        ("compute_type", c_int32), ("round_output_bf16", c_int32),
        ("relu_mask", c_void_p), ("ld_mask", c_int32), ("stride_mask", c_int64),
    ]


class PpoLossArgs(Structure):
    _fields_ = [
        ("mu", c_void_p), ("mu_stride", c_int64), ("value", c_void_p), ("value_stride", c_int64), ("logstd", c_void_p),
        ("idx", c_void_p), ("actions", c_void_p), ("actions_stride", c_int64), ("old_mu", c_void_p), ("old_mu_stride", c_int64),
        ("old_logstd", c_void_p), ("old_neglogp", c_void_p), ("advantages", c_void_p), ("old_values", c_void_p), ("returns", c_void_p),
        ("rows", c_int32), ("num_actions", c_int32),
        ("e_clip", c_float), ("critic_coef", c_float), ("bounds_loss_coef", c_float), ("clip_value", c_int32), ("has_bounds_loss", c_int32),
        ("dmu", c_void_p), ("dmu_stride", c_int64), ("dvalue", c_void_p), ("dvalue_stride", c_int64),
        ("partials", c_void_p), ("num_blocks", c_int32),
        ("dmu16", c_void_p), ("dmu16_stride", c_int64), ("dvalue16", c_void_p), ("dvalue16_stride", c_int64),
    ]


GEMM_RED_CONTIG, GEMM_OUT_CONTIG = 0, 1
GEMM_COMPUTE_F32, GEMM_COMPUTE_BF16, GEMM_COMPUTE_F32X3 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_SILU, ACT_SILU_D = 0, 1, 2, 3          # SILU_D: C2 receives d silu / d z instead of z (backward: EPI_MUL_AUX)
EPI_BIAS_ACT, EPI_RELU_GRAD, EPI_SILU_GRAD, EPI_MUL_AUX = 0, 1, 2, 3

P = c_void_p  # every device pointer crosses the ABI as void*
c_double = ctypes.c_double

# name -> (restype, argtypes); must list EVERY symbol of include/pulse_hip.h
SIGNATURES = {
    "pulse_abi_version": (c_int, []),
    "pulse_last_error": (c_char_p, []),
    "pulse_quat_mul": (c_int, [P, P, P, c_int64, P]),
    "pulse_quat_conjugate": (c_int, [P, P, c_int64, P]),
    "pulse_quat_rotate": (c_int, [P, P, P, c_int64, P]),
    "pulse_quat_to_angle_axis": (c_int, [P, P, P, c_int64, P]),
    "pulse_quat_to_exp_map": (c_int, [P, P, c_int64, P]),
    "pulse_quat_to_tan_norm": (c_int, [P, P, c_int64, P]),
    "pulse_exp_map_to_quat": (c_int, [P, P, c_int64, P]),
    "pulse_slerp": (c_int, [P, P, P, P, c_int64, P]),
    "pulse_calc_heading": (c_int, [P, P, c_int64, P]),
    "pulse_calc_heading_quat": (c_int, [P, P, c_int64, c_int, P]),
    "pulse_sizeof_im_step_args": (c_int, []),
    "pulse_self_obs_width": (c_int, [c_int, c_int]),
    "pulse_task_obs_width": (c_int, [c_int, c_int, c_int]),
    "pulse_self_obs_width_ex": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "pulse_im_step": (c_int, [POINTER(ImStepArgs), P]),
    "pulse_sizeof_task_step_args": (c_int, []),
    "pulse_task_obs_size": (c_int, [c_int]),
    "pulse_task_step": (c_int, [POINTER(TaskStepArgs), P]),
    "pulse_gemm_set_option": (c_int, [c_int, c_int]),
    "pulse_gemm_set_debug_buffer": (c_int, [P]),
    "pulse_gemm_last_tile": (c_int, []),
    "pulse_gemm_x3_mode": (c_int, []),
    "pulse_sizeof_amp_obs_args": (c_int, []),
    "pulse_amp_obs_width": (c_int, [c_int, c_int, c_int]),
    "pulse_amp_obs": (c_int, [POINTER(AmpObsArgs), P]),
    "pulse_sizeof_amp_hist_args": (c_int, []),
    "pulse_amp_hist_init": (c_int, [POINTER(AmpHistArgs), P]),
    "pulse_sizeof_rollout_record_args": (c_int, []),
    "pulse_rollout_record": (c_int, [POINTER(RolloutRecordArgs), P]),
    "pulse_rollout_meters": (c_int, [P, c_int32, c_int32, P, P, c_float, P]),
    "pulse_kinematic_sim_step": (c_int, [P, P, P, c_int64, c_int32, P, P, P, P, P, P, P, P, c_int32, P]),
    "pulse_sizeof_pd_sim_args": (c_int, []),
    "pulse_pd_sim_step": (c_int, [POINTER(PdSimArgs), P]),
    "pulse_sizeof_motion_state_args": (c_int, []),
    "pulse_motion_state": (c_int, [POINTER(MotionStateArgs), P]),
    "pulse_gae": (c_int, [P, P, P, P, c_int32, c_int32, c_int64, c_int64, c_float, c_float, P, P, P]),
    "pulse_sizeof_gemm_desc": (c_int, []),
    "pulse_gemm_f32": (c_int, [POINTER(GemmDesc), P]),
    "pulse_gemm_x3p": (c_int, [POINTER(GemmX3pDesc), P]),
    "pulse_gemm_x3p_row_tiles": (c_int, [c_int32, c_int32, c_int32]),
    "pulse_traj_step": (c_int, [POINTER(TrajStepArgs), P]),
    "pulse_sizeof_traj_step_args": (c_int, []),
    "pulse_traj_generate": (c_int, [POINTER(TrajGenArgs), P]),
    "pulse_vae_embed": (c_int, [POINTER(VaeEmbedArgs), P]),
    "pulse_vae_kin_loss": (c_int, [POINTER(VaeKinArgs), P]),
    "pulse_vae_head_backward": (c_int, [POINTER(VaeHeadBwdArgs), P]),
    "pulse_sizeof_vae_embed_args": (c_int, []),
    "pulse_sizeof_vae_kin_args": (c_int, []),
    "pulse_sizeof_vae_head_bwd_args": (c_int, []),
    "pulse_sizeof_gemm_x3p_desc": (c_int, []),
    "pulse_split_planes": (c_int, [P, c_int64, c_int32, c_int32, P, c_int64, c_int32, c_int32, P, P]),
    "pulse_reduce_slabs": (c_int, [P, c_int32, c_int64, c_int64, P, c_float, P]),
    "pulse_colsum_partial": (c_int, [P, c_int32, c_int32, c_int32, c_int32, P, c_int64, P]),
    "pulse_rms_normalize": (c_int, [P, c_int64, P, c_int32, c_int32, P, P, c_float, c_float, c_int32, P, c_int64, c_int32, P, c_int32, P]),
    "pulse_rms_normalize_copy": (c_int, [P, c_int64, P, c_int32, c_int32, P, P, c_float, c_float, P, c_int64, c_int32, P, c_int32, P, c_int64, P]),
    "pulse_rms_normalize_planes": (c_int, [P, c_int64, P, c_int32, c_int32, P, P, c_float, c_float, P, c_int64, c_int32, P, c_int32, P, c_int64, c_int64, P]),
    "pulse_rms_normalize_b16": (c_int, [P, c_int64, P, c_int32, c_int32, P, P, c_float, c_float, P, c_int64, c_int32, P, c_int32, P]),
    "pulse_transpose_to_b16": (c_int, [P, c_int64, c_int32, c_int32, P, c_int64, c_int32, c_int64, c_int64, P]),
    "pulse_colsum_partial_b16": (c_int, [P, c_int32, c_int32, c_int64, c_int32, P, c_int64, P]),
    "pulse_colsum_weighted_b16": (c_int, [P, c_int32, c_int32, c_int64, P, c_int64, c_int32, P, c_int64, P]),
    "pulse_disc_penalty": (c_int, [P, c_int64, c_int32, c_int32, c_float, P, c_int64, P, c_int64, P, c_int32, P]),
    "pulse_disc_reg": (c_int, [P, P, c_int32, POINTER(c_int64), POINTER(c_int64), POINTER(c_float), P, c_int32, P]),
    "pulse_sizeof_b16_transpose": (c_int, []),
    "pulse_weights_to_b16": (c_int, [P, c_int64, P, c_int32, POINTER(B16Transpose), P]),
    "pulse_reduce_grads": (c_int, [P, c_int64, c_int32, POINTER(c_int64), POINTER(c_int64), POINTER(c_int32), POINTER(c_float), POINTER(ctypes.c_void_p),
                                   POINTER(c_int64), P, c_float, P, P, P, c_int32, P]),
    "pulse_disc_reward": (c_int, [P, c_int64, c_int64, c_float, P, c_int64, P]),
    "pulse_disc_head_b16": (c_int, [P, c_int64, c_int32, c_float, P, c_int64, P, c_int64, P, P, P]),
    "pulse_rms_update": (c_int, [P, P, P, P, c_int32, c_int32, c_double, c_double, P]),
    "pulse_policy_sample": (c_int, [P, c_int64, P, P, c_int64, P, c_int64, P, P, c_int32, c_int32, P, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_int64, P]),
    "pulse_sizeof_ppo_loss_args": (c_int, []),
    "pulse_ppo_loss": (c_int, [POINTER(PpoLossArgs), P]),
    "pulse_advantage_moments": (c_int, [P, P, c_int64, P, P, c_int32, P]),
    "pulse_advantage_normalize": (c_int, [P, c_int64, P, c_int32, P]),
    "pulse_sqnorm_partial": (c_int, [P, c_int64, P, c_int32, P]),
    "pulse_disc_head": (c_int, [P, c_int64, c_int32, c_float, P, c_int64, P, P]),
    "pulse_adam_step": (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int32, c_float, P, c_int32, P, P]),
    "pulse_adam_step_multi": (c_int, [c_int32, POINTER(ctypes.c_void_p), POINTER(ctypes.c_void_p), POINTER(ctypes.c_void_p), POINTER(ctypes.c_void_p),
                                      POINTER(c_int64), c_float, c_float, c_float, c_float, c_float, c_int32, c_float, P, c_int32, P, P]),
}

_lib = None
_load_error = None


def load():
    """Load (once) and return the ctypes library handle; raise loudly if unavailable."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:
        raise _load_error
    try:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} does not exist")
        # PyTorch ships its own libamdhip64; it must be the HIP runtime of the process (the streams and device pointers handed to
        # this library are torch's).  Loading libpulse_hip.so first would pull in the system ROCm runtime instead and every launch
        # would then fail with "no ROCm-capable device is detected".
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        v = lib.pulse_abi_version()
        if v != ABI_VERSION:
            raise OSError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}")
    except (OSError, AttributeError) as exc:
        _load_error = PulseLibraryError(
            "pulse_amd: the HIP extension libpulse_hip.so is required and could not be loaded "
            f"({exc}). There is no CPU fallback. Build it with `python pulse_amd/csrc/build.py`.")
        raise _load_error from exc
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().pulse_last_error()
        raise PulseLibraryError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")
