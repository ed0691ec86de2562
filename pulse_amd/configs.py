"""The BASELINE.json configurations as concrete configs (values from the reference's YAML files).

learning/im.yaml:46-91 hyper-parameters; env/env_im.yaml env switches.  ``make_agent`` wires a
recorded synthetic rollout (pulse_amd/env/sim.py) to HumanoidIm and a CommonAgent.
"""
import copy

NETWORK_IM = {            # phc/data/cfg/learning/im.yaml:12-44
    "name": "amp", "separate": True,
    "space": {"continuous": {"mu_activation": "None", "sigma_activation": "None", "mu_init": {"name": "default"},
                             "sigma_init": {"name": "const_initializer", "val": -2.9}, "fixed_sigma": True, "learn_sigma": False}},
    "mlp": {"units": [1024, 512], "activation": "relu", "d2rl": False, "initializer": {"name": "default"}},
    "disc": {"units": [1024, 512], "activation": "relu", "initializer": {"name": "default"}},
}

PPO_IM = {                # phc/data/cfg/learning/im.yaml:46-91
    "name": "Humanoid", "multi_gpu": False, "ppo": True, "mixed_precision": False, "normalize_input": True,
    "normalize_value": True, "reward_shaper": {"scale_value": 1}, "normalize_advantage": True, "gamma": 0.99, "tau": 0.95,
    "learning_rate": 2e-5, "lr_schedule": "constant", "entropy_coef": 0.0, "truncate_grads": True, "grad_norm": 50.0,
    "e_clip": 0.2, "horizon_length": 32, "minibatch_size": 16384, "mini_epochs": 6, "critic_coef": 5, "clip_value": False,
    "bounds_loss_coef": 10,
    # AMP (learning/im.yaml:78-91)
    "amp_obs_demo_buffer_size": 200000, "amp_replay_buffer_size": 200000, "amp_replay_keep_prob": 0.01, "amp_batch_size": 512,
    "amp_minibatch_size": 4096, "disc_coef": 5, "disc_logit_reg": 0.01, "disc_grad_penalty": 5, "disc_reward_scale": 2,
    "disc_weight_decay": 0.0001, "normalize_amp_input": True, "task_reward_w": 0.5, "disc_reward_w": 0.5,
}

ENV_IM = {"obs_v": 6, "self_obs_v": 1, "power_reward": True, "local_root_obs": True, "root_height_obs": True,
          "enableEarlyTermination": True, "terminationDistance": 0.25, "episode_length": 300}

NETWORK_Z = {             # phc/data/cfg/learning/im_z_fit.yaml:12-50 (network: amp_z)
    "name": "amp_z", "separate": True,
    "space": {"continuous": {"mu_activation": "None", "sigma_activation": "None", "mu_init": {"name": "default"},
                             "sigma_init": {"name": "const_initializer", "val": -2.9}, "fixed_sigma": True, "learn_sigma": False}},
    "mlp": {"units": [3096, 2048, 1024], "activation": "silu", "d2rl": False, "initializer": {"name": "default"}},
    "task_mlp": {"units": [1536, 1024, 512], "activation": "silu", "d2rl": False, "initializer": {"name": "default"}},
}

ENV_IM_VAE = dict(ENV_IM, **{   # phc/data/cfg/env/env_im_vae.yaml:19-55 (PULSE distillation)
    "embedding_norm": 1, "embedding_size": 32, "z_type": "vae", "use_vae_prior": True, "use_ar1_prior": True,
    "use_vae_clamped_prior": True, "vae_var_clamp_max": 2, "kld_coefficient": 0.01, "kld_coefficient_min": 0.001, "kld_anneal": True,
    "ar1_coefficient": 0.005, "only_kin_loss": True, "distill": True, "save_kin_info": True, "cycle_motion": True})

NETWORK_Z_READER = {      # phc/data/cfg/learning/pulse_z_task.yaml:10-44 (network: amp_z_reader)
    "name": "amp_z_reader", "separate": True,
    "space": {"continuous": {"mu_activation": "None", "sigma_activation": "None", "mu_init": {"name": "default"},
                             "sigma_init": {"name": "const_initializer", "val": -1.0}, "fixed_sigma": True, "learn_sigma": False}},
    "mlp": {"units": [2048, 1024, 512], "activation": "silu", "d2rl": False, "initializer": {"name": "default"}},
}

NETWORK_SEPT = {          # phc/data/cfg/learning/pulse_z_terrain.yaml:12-51 (network: amp_sept)
    "name": "amp_sept", "separate": True,
    "space": {"continuous": {"mu_activation": "None", "sigma_activation": "None", "mu_init": {"name": "default"},
                             "sigma_init": {"name": "const_initializer", "val": -1.0}, "fixed_sigma": True, "learn_sigma": False}},
    "mlp": {"units": [2048, 1024, 512], "activation": "silu", "d2rl": False, "initializer": {"name": "default"}},
    "task_mlp": {"units": [512, 256], "activation": "silu", "d2rl": False, "initializer": {"name": "default"}},
}

ENV_TERRAIN_Z = {"local_root_obs": True, "root_height_obs": True, "enableEarlyTermination": True, "episode_length": 300, "enableTaskObs": True,
                 "numTrajSamples": 10, "trajSampleTimestep": 0.5, "speedMin": 0.0, "speedMax": 3.0, "accelMax": 2.0, "sharpTurnProb": 0.02,
                 "terrain_obs": True, "terrain_obs_type": "square", "terrain_obs_root": "head", "use_center_height": True, "power_reward": False,
                 "terrain": {"terrainType": "trimesh"}, "embedding_size": 32, "z_type": "vae"}   # phc/data/cfg/env/env_pulse_terrain.yaml

ENV_SPEED_Z = {"local_root_obs": True, "root_height_obs": True, "enableEarlyTermination": True, "episode_length": 300, "enableTaskObs": True,
               "tarSpeedMin": 0.0, "tarSpeedMax": 5.0, "speedChangeStepsMin": 100, "speedChangeStepsMax": 200, "power_reward": True,
               "embedding_size": 32, "z_type": "vae"}     # phc/data/cfg/env/env_pulse_amp.yaml (HumanoidSpeedZ)

CONFIGS = {
    # BASELINE.json configs[0]: 64-env synthetic rollout (horizon 16), 2x512 MLP, one PPO+GAE epoch
    "cfg1": {"num_envs": 64, "horizon_length": 16, "minibatch_size": 256, "units": [512, 512]},
    # BASELINE.json configs[1]: 4096 SMPL humanoids, horizon 32, imitation reward/obs + PPO
    "cfg2": {"num_envs": 4096, "horizon_length": 32, "minibatch_size": 16384, "units": [1024, 512]},
    # BASELINE.json configs[2]: 8192 envs, PULSE VAE encoder/decoder in the policy head (latent 32); kin/VAE loss (PULSE training)
    "cfg3": {"num_envs": 8192, "horizon_length": 32, "minibatch_size": 16384, "network": "amp_z", "env": "vae", "agent": "amp",
             "extra": {"use_seq_rl": True}},
    # same network trained with the PPO loss (SURVEY.md 8d cfg 3, first variant)
    "cfg3_ppo": {"num_envs": 8192, "horizon_length": 32, "minibatch_size": 16384, "network": "amp_z", "env": "vae_ppo", "agent": "amp"},
    # BASELINE.json configs[4]: AMP discriminator + PPO, 8192 envs, bf16: training GEMMs on the bf16 MFMA over fp32 master weights
    "cfg5": {"num_envs": 8192, "horizon_length": 32, "minibatch_size": 16384, "units": [1024, 512], "env": "amp", "agent": "amp",
             "extra": {"enable_disc": True, "mixed_precision": True}},
    "cfg5_f32": {"num_envs": 8192, "horizon_length": 32, "minibatch_size": 16384, "units": [1024, 512], "env": "amp", "agent": "amp",
                 "extra": {"enable_disc": True}},
    "cfg5_small": {"num_envs": 64, "horizon_length": 16, "minibatch_size": 256, "units": [512, 512], "env": "amp", "agent": "amp",
                   "extra": {"enable_disc": True, "amp_minibatch_size": 64, "amp_obs_demo_buffer_size": 4096, "amp_replay_buffer_size": 4096,
                             "amp_batch_size": 128}},
    # downstream task on a frozen PULSE decoder: HumanoidSpeedZ, policy = amp_z_reader (learning=pulse_z_task), latent action 32
    "speed_z": {"num_envs": 4096, "horizon_length": 32, "minibatch_size": 16384, "network": "amp_z_reader", "env": "speed_z", "agent": "amp"},
    "speed_z_small": {"num_envs": 64, "horizon_length": 16, "minibatch_size": 256, "network": "amp_z_reader", "env": "speed_z", "agent": "amp",
                      "units": [256, 128]},
    # terrain traversal on a frozen PULSE decoder: HumanoidPedestrianTerrainZ, policy = amp_sept (learning=pulse_z_terrain; env_pulse_terrain.yaml: 1536 envs)
    "terrain_z": {"num_envs": 1536, "horizon_length": 32, "minibatch_size": 16384, "network": "amp_sept", "env": "terrain_z", "agent": "amp"},
    "terrain_z_small": {"num_envs": 64, "horizon_length": 16, "minibatch_size": 256, "network": "amp_sept", "env": "terrain_z", "agent": "amp",
                        "units": [256, 128]},
    # small shapes of the same graphs for tests
    "cfg3_small": {"num_envs": 64, "horizon_length": 16, "minibatch_size": 256, "network": "amp_z", "env": "vae", "agent": "amp",
                   "extra": {"use_seq_rl": True}},
    "cfg3_ppo_small": {"num_envs": 64, "horizon_length": 16, "minibatch_size": 256, "network": "amp_z", "env": "vae_ppo", "agent": "amp"},
}


def agent_config(name, **overrides):
    c = CONFIGS[name]
    if c.get("network") == "amp_z":
        net = copy.deepcopy(NETWORK_Z)
    elif c.get("network") == "amp_sept":
        net = copy.deepcopy(NETWORK_SEPT)
        if "units" in c:
            net["mlp"]["units"] = list(c["units"])
    elif c.get("network") == "amp_z_reader":
        net = copy.deepcopy(NETWORK_Z_READER)
        if "units" in c:
            net["mlp"]["units"] = list(c["units"])
    else:
        net = copy.deepcopy(NETWORK_IM)
        net["mlp"]["units"] = list(c["units"])
    cfg = copy.deepcopy(PPO_IM)
    cfg.update({"horizon_length": c["horizon_length"], "minibatch_size": c["minibatch_size"], "network": net})
    cfg.update(c.get("extra", {}))
    cfg.update(overrides)
    cfg["_env_kind"], cfg["_agent_kind"] = c.get("env", "im"), c.get("agent", "common")
    return cfg, c["num_envs"]


def make_env(num_envs, horizon, device, seed=1234, rank=0, rollout=None, env_kind="im", reference="recorded", env_overrides=None):
    """``reference``: 'recorded' = pre-recorded rigid-body / reference frames (RecordedRollout, what the CPU oracle agent replays);
    'motion_lib' = reference motion queried from the HBM-resident MotionLib every step, physics stand-in tracking it."""
    from .env.humanoid_im import HumanoidIm, VecTaskPythonWrapper
    if env_kind in ("speed_z", "reach_z", "strike_z", "terrain_z"):
        # HumanoidSpeedZ & co: synthetic task physics, the frozen decoder initialised from a (random-init) PULSE checkpoint
        import torch
        from .env import humanoid_tasks as HT
        from .learning.network_z import AMPZNetwork
        env_cfg = dict(ENV_TERRAIN_Z if env_kind == "terrain_z" else ENV_SPEED_Z)
        env_cfg.update(env_overrides or {})
        # the terrain task's humanoids walk inside the synthetic height field (cells of 0.1 m): start them near its middle
        sim = HT.SyntheticTaskSim(num_envs, horizon + 1, device, seed=seed, rank=rank, xy_offset=(13.0, 15.0) if env_kind == "terrain_z" else (0.0, 0.0))
        cls = {"speed_z": HT.HumanoidSpeedZ, "reach_z": HT.HumanoidReachZ, "strike_z": HT.HumanoidStrikeZ,
               "terrain_z": HT.HumanoidPedestrianTerrainZ}[env_kind]
        task = cls({"env": env_cfg}, sim, device=device)
        znet = AMPZNetwork(NETWORK_Z, actions_num=69, self_obs_size=task.get_self_obs_size(), task_obs_size=576,
                           task_obs_size_detail={"embedding_size": 32, "z_type": "vae", "use_vae_prior": True, "use_vae_clamped_prior": True,
                                                 "vae_var_clamp_max": 2}, device=device)
        rms = {"running_mean": torch.zeros(934, dtype=torch.float64), "running_var": torch.ones(934, dtype=torch.float64)}
        task.initialize_z_models({"model": znet.state_dict(), "running_mean_std": rms}, NETWORK_Z)
        return VecTaskPythonWrapper(task, rl_device=device), None
    env_cfg = dict(ENV_IM_VAE) if env_kind in ("vae", "vae_ppo") else dict(ENV_IM)
    if env_kind == "vae_ppo":
        env_cfg.update({"only_kin_loss": False, "save_kin_info": False, "distill": False})
    if env_kind == "amp":
        env_cfg.update({"enable_amp_obs": True, "numAMPObsSteps": 10})
    env_cfg.update(env_overrides or {})
    if reference == "motion_lib":
        from . import synthetic as syn
        from .env.motion_lib import MotionLib
        from .env.sim import KinematicSim, PdSim
        tables = syn.synthetic_motion_library(syn.make_generator(seed + 5, rank), min(num_envs, 1024))
        motion = MotionLib.from_tables(tables, device)
        sim_cls = PdSim if env_cfg.pop("physics", "tracking") == "pd" else KinematicSim        # "pd": action-dependent stand-in
        sim = sim_cls(num_envs, horizon + 1, device, seed=seed, rank=rank)
        task = HumanoidIm({"env": env_cfg}, sim, motion, device=device)
        return VecTaskPythonWrapper(task, rl_device=device), None
    from .env.sim import RecordedMotion, RecordedRollout, RecordedSim
    if rollout is None:
        rollout = RecordedRollout(num_envs, horizon + 1, seed=seed, rank=rank)
    rollout.to(device)
    sim = RecordedSim(rollout)
    motion = RecordedMotion(rollout, sim)
    task = HumanoidIm({"env": env_cfg}, sim, motion, device=device)
    task.progress_buf.copy_(rollout.init_progress)
    return VecTaskPythonWrapper(task, rl_device=device), rollout


def make_agent(name="cfg2", device="cuda:0", seed=1234, rank=0, rollout=None, reference="recorded", env_overrides=None, **overrides):
    from .learning.amp_agent import AMPAgent
    from .learning.common_agent import CommonAgent
    num_envs_override = overrides.pop("num_envs_override", None)
    cfg, num_envs = agent_config(name, **overrides)
    if num_envs_override is not None:
        num_envs = int(num_envs_override)
    vec_env, rollout = make_env(num_envs, cfg["horizon_length"], device, seed=seed, rank=rank, rollout=rollout, env_kind=cfg["_env_kind"],
                                reference=reference, env_overrides=env_overrides)
    cfg.update({"vec_env": vec_env, "device": device, "seed": seed})
    cls = AMPAgent if cfg["_agent_kind"] == "amp" else CommonAgent
    return cls("pulse_amd", cfg), rollout
