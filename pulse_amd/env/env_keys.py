"""Every ``cfg["env"]`` key the reference's task classes read, and what this implementation does with it.

The reference reads its env options with ``cfg["env"].get(key, default)`` scattered over the task class chain
(phc/env/tasks/humanoid.py:245-350, humanoid_amp.py:76-135, humanoid_im.py:30-100, 496-527, humanoid_z.py:20-80, humanoid_im_getup.py,
humanoid_im_distill.py, humanoid_speed.py / _reach.py / _strike.py / _traj.py / _pedestrian_terrain.py, base_task.py:60-131).  A key that is
silently ignored here would change what a shipped configuration computes without anybody noticing (round 5 shipped exactly that hole:
``zero_out_far``).  ``audit`` therefore sorts EVERY key of an env dict into one of three classes:

  HONOURED   this package reads the key and does what the reference does with it;
  INERT      the key configures a subsystem that is out of scope by contract (Isaac Gym scene / asset creation, the viewer, AMASS loading)
             or is read by the reference into an attribute nothing consumes; accepted and ignored, with the reason recorded below;
  UNBUILT    the reference acts on the key, this package does not: accepted ONLY at the value(s) for which the reference's behaviour is the
             one built here, anything else raises NotImplementedError naming the key and the reference lines that implement it.

A key in none of the three is one the reference does not read on this path either (legacy spellings, notes): ignoring it IS the
reference's behaviour; it is reported once through ``warnings`` so typos are visible.  tests/test_env_keys.py re-derives the set of keys the
reference reads from its sources (when /root/reference is mounted) and fails if one is missing here, and walks every shipped
``phc/data/cfg/env/*.yaml``.
"""
import warnings

# ---------------------------------------------------------------------------------------------------------------------------------------
HONOURED = {
    # observation / reward / reset switches of the imitation task
    "obs_v", "self_obs_v", "past_track_steps", "force_sensor_joints", "fut_tracks", "numTrajSamples", "trajSampleTimestepInv",
    "local_root_obs", "root_height_obs", "full_body_reward", "power_reward", "power_coefficient", "reward_specs",
    "enableEarlyTermination", "episode_length", "cycle_motion", "trackBodies", "reset_bodies", "terminationDistance", "stateInit",
    "controlFrequencyInv", "strict_eval", "cycle_motion_xp", "fut_tracks_dropout", "add_obs_noise", "res_action",
    "occl_training", "occl_training_prob",
    # zero_out_far (humanoid.py:311-329)
    "zero_out_far", "zero_out_far_train", "zero_out_far_steps", "close_distance", "far_distance",
    # robot switches the env dict may carry (robot/*.yaml merged by the caller)
    "has_upright_start", "has_dof_subset", "has_shape_obs", "has_weight_obs",
    # AMP observations
    "enable_amp_obs", "numAMPObsSteps", "ampRootHeightObs", "key_bodies", "add_amp_input_noise",
    # PULSE / distillation attributes the agent reads off the task (amp_agent.py:59-63, 773-832)
    "temp_running_mean", "kin_lr", "save_kin_info", "only_kin_loss", "distill", "z_type", "kld_coefficient", "kld_coefficient_min",
    "ar1_coefficient", "kld_anneal", "use_ar1_prior", "use_vae_prior", "use_vae_prior_regu", "embedding_size", "embedding_norm",
    "use_vae_clamped_prior", "vae_var_clamp_max", "proj_norm", "auto_pmcp", "auto_pmcp_soft", "shape_resampling_interval", "fitting", "models",
    # get-up task (humanoid_im_getup.py:42-66)
    "recoveryEpisodeProb", "recoverySteps", "fallInitProb", "getup_schedule", "getup_udpate_epoch",
    # downstream tasks (humanoid_speed.py / _reach.py / _strike.py / _traj.py / _pedestrian_terrain.py)
    "enableTaskObs", "tarSpeedMin", "tarSpeedMax", "speedChangeStepsMin", "speedChangeStepsMax", "tarChangeStepsMin", "tarChangeStepsMax",
    "tarDistMax", "tarHeightMin", "tarHeightMax", "reachBodyName", "strikeBodyNames", "terminationHeight", "contact_bodies",
    "trajSampleTimestep", "speedMin", "speedMax", "accelMax", "sharpTurnProb", "sensor_extent", "sensor_res", "fuzzy_target",
    "terrain", "terrain_obs", "terrain_obs_type", "terrain_obs_root", "use_center_height",
    "power_usage_reward", "power_usage_coefficient",      # read by the speed / strike tasks only (humanoid_speed.py:43-53, humanoid_strike.py:37-40)
    # keys of THIS package (no reference counterpart): seeds of the synthetic stand-ins, stand-in selection
    "motion_clock_seed", "obs_noise_seed", "occl_seed", "shape_seed", "task_seed", "getup_seed", "physics", "contactBodies", "tarDistMin", "nearDist", "nearProb",
}

_SIM = "Isaac Gym scene / actor / asset creation (closed-source physics: out of scope, SURVEY.md section 2 #20, #36)"
_LOAD = "consumed while motions are loaded (motion_lib_smpl / resample_motions: AMASS + smpl_sim, out of scope): the caller hands over a built MotionLib"
_VIEW = "viewer / debug drawing (out of scope)"
_DEAD = "the reference stores it in an attribute that nothing on the training path reads"
_TEACH = "passed through get_task_obs_size_detail to the teacher / PNN network builders (learning/teacher.py takes its structure as arguments)"

INERT = {
    "task": "selects the task class (utils/parse_task.py:57-70): the caller instantiates the class",
    "project_name": "run metadata", "notes": "run metadata",
    "num_envs": "taken from the injected simulator (checked against it when given)",
    "env_spacing": _SIM, "plane": _SIM, "kp_scale": _SIM, "kd_scale": _SIM, "power_scale": _SIM, "pd_control": _SIM,
    "default_humanoid_mass": _SIM, "numActions": _SIM, "numObservations": _SIM, "numStates": _SIM, "asset": _SIM,
    "enable_debug_vis": _VIEW, "show_sensors": _VIEW, "is_flag_run": _DEAD,
    "motion_file": _LOAD, "min_length": _LOAD, "max_len": _LOAD, "seq_motions": _LOAD, "hard_negative": _LOAD,
    "eval_full": _DEAD, "kin_policy": _DEAD, "partial_running_mean": _DEAD, "vae_reader": _DEAD, "z_model": _DEAD, "z_read": _DEAD,
    "z_uniform": _DEAD, "use_vae_prior_loss": _DEAD, "velocity_map": _DEAD, "tarSpeed": _DEAD,
    "num_prim": _TEACH, "training_prim": _TEACH, "actors_to_load": _TEACH, "has_lateral": _TEACH,
    "distill_model_config": "structure of the frozen PULSE networks: HumanoidZ.initialize_z_models takes the checkpoint and network params as arguments",
    "hybridInitProb": "only read when stateInit is Hybrid (humanoid_amp.py:490-505), which raises here",
    "dict_size": "VQ dictionary size: only read for z_type vq_vae variants, which raise here",
    "embedding_partion": "VQ partition count: only read for z_type vq_vae variants, which raise here",
    "vae_prior_fixed_logvar": "only read with use_vae_fixed_prior, which raises here",
    "num_env_group": "only read with divide_group, which raises here",
    "small_terrain": "debug switch of the terrain mesh creation (" + _SIM + ")",
}

# key -> (values at which the reference's behaviour is what is built here, where the reference implements the rest)
UNBUILT = {
    "addInputNoise": ((False,), "vec_task / task input noise"),
    "remove_disc_rot": ((False,), "humanoid.py:413-416 (discriminator dof subset without global rotation)"),
    "amp_obs_v": ((1,), "humanoid_amp.py:300-314, 670-680 (build_amp_observations_smpl_v2)"),
    "numAMPEncObsSteps": ("==numAMPObsSteps", "humanoid_amp.py:94, 834-880 (CALM encoder windows)"),
    "enableHistObs": ((False,), "humanoid_amp.py:320, 509-517"),
    "is_discrete": ((False,), "base_task.py:90; amp_agent.py:42-44 (discrete action heads)"),
    "control_mode": (("isaac_pd",), "humanoid.py:1250-1290 (torques computed in Python for control_mode pd)"),
    "divide_group": ((False,), "humanoid_pedestrian_terrain.py:265-289, 431-468, 744-766 (crowd observations)"),
    "group_obs": ((False,), "humanoid_pedestrian_terrain.py:431-437, 744-766 (crowd observations)"),
    "disable_group_obs": ((False,), "humanoid_pedestrian_terrain.py:749"),
    "z_readout": ((False,), "amp_network_z_builder.py:63, amp_network_z_reader_builder.py:36"),
    "z_all": ((False,), "amp_network_z_builder.py:37 (z fed to every decoder layer)"),
    "vae_prior_policy": ((False,), "amp_network_z_reader_builder.py:38-57"),
    "use_vae_fixed_prior": ((False,), "amp_network_z_builder.py:102, 237, 529; amp_agent.py:785"),
    "use_vae_sphere_prior": ((False,), "amp_network_z_builder.py:238"),
    "use_vae_sphere_posterior": ((False,), "amp_network_z_builder.py:118; humanoid_z.py:106"),
    "distill_z_model": ((False,), "humanoid_im_distill.py:43-63, 186-231 (a PULSE model as the distillation teacher)"),
}

ALL_KNOWN = HONOURED | set(INERT) | set(UNBUILT)
_warned = set()


def _inactive(key, value, env):
    ok, _ = UNBUILT[key]
    if ok == "==numAMPObsSteps":
        return int(value) == int(env.get("numAMPObsSteps", 10))
    return any(value == v for v in ok)


def audit(env, where="HumanoidIm"):
    """Raise NotImplementedError if ``env`` switches on behaviour of the reference that is not built; warn (once per key) about keys
    the reference does not read on this path."""
    bad = [k for k in env if k in UNBUILT and not _inactive(k, env[k], env)]
    if bad:
        lines = [f"  {k} = {env[k]!r}: not built (accepted: {UNBUILT[k][0]}); reference: phc/.../{UNBUILT[k][1]}" for k in bad]
        raise NotImplementedError(f"{where}: env option(s) the reference acts on but this implementation does not:\n" + "\n".join(lines))
    unknown = sorted(k for k in env if k not in ALL_KNOWN and k not in _warned)
    if unknown:
        _warned.update(unknown)
        warnings.warn(f"{where}: env key(s) {unknown} are not read by the reference's task classes on this path either; ignored", stacklevel=3)
