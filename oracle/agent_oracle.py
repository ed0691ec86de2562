"""PyTorch-CPU restatement of the learner path: CommonAgent + the rl_games pieces it inherits.

ORACLE / TEST INFRASTRUCTURE -- never imported by ``pulse_amd``.  Used by tests/ (parity on
identical recorded rollouts) and by bench.py's ``cpu_baseline`` leg (kind "port": the reference's
agent cannot be imported here because rl_games / isaacgym / gym are absent).

Everything is plain eager PyTorch: nn.Linear modules, autograd, torch.optim.Adam,
nn.utils.clip_grad_norm_, the reference's time-major (T, N, .) experience buffer and its
transposing ``swap_and_flatten01``.  Line references are to /root/reference.

  OracleRunningMeanStd   phc/utils/running_mean_std.py:9-109   (pinned: tests/golden/rms.npz)
  OracleNet              network_builder.py:188-291 + amp_network_builder.py:19-40,127-148,206-211 (MLP branch),
                         ModelA2CContinuousLogStd (rl_games 1.1.4) -- UNPINNED 3P boundary
  OracleEnv              HumanoidIm.step over a recorded rollout (humanoid.py:1315-1331 order)
  OracleCommonAgent      phc/learning/common_agent.py:36-98,191-599 (GAE / losses pinned: tests/golden/agent_math.npz)
"""
import math
import time

import torch
import torch.nn as nn

from . import env_oracle as E


class OracleRunningMeanStd(nn.Module):
    """phc/utils/running_mean_std.py (non-per-channel branch)."""

    def __init__(self, insize, epsilon=1e-05):
        super().__init__()
        self.insize = insize
        self.mean_size = insize[0]
        self.epsilon = epsilon
        self.register_buffer("running_mean", torch.zeros(insize, dtype=torch.float64))
        self.register_buffer("running_var", torch.ones(insize, dtype=torch.float64))
        self.register_buffer("count", torch.ones((), dtype=torch.float64))
        self.forzen = False

    def freeze(self):
        self.forzen = True

    def forward(self, input, unnorm=False):
        mean, var = self.running_mean, self.running_var
        if unnorm:
            y = torch.clamp(input, min=-5.0, max=5.0)
            y = torch.sqrt(var.float() + self.epsilon) * y + mean.float()
        else:
            y = (input - mean.float()) / torch.sqrt(var.float() + self.epsilon)
            y = torch.clamp(y, min=-5.0, max=5.0)
        if self.training and not self.forzen:
            bm, bv, bc = input.mean([0]), input.var([0]), input.size()[0]
            delta = bm - mean
            tot = self.count + bc
            new_mean = mean + delta * bc / tot
            m2 = var * self.count + bv * bc + delta ** 2 * self.count * bc / tot
            self.running_mean, self.running_var, self.count = new_mean, m2 / tot, tot
        return y


class OracleNet(nn.Module):
    """AMPBuilder.Network (MLP branch, no discriminator) + the Gaussian log-std model wrapper."""

    def __init__(self, obs_dim, actions_num, units, activation="relu", sigma_val=-2.9):
        super().__init__()
        act = {"relu": nn.ReLU, "silu": nn.SiLU}[activation]

        def mlp():
            layers, i = [], obs_dim
            for u in units:
                layers += [nn.Linear(i, u), act()]
                i = u
            return nn.Sequential(*layers)
        # construction order of A2CBuilder.Network.__init__ (network_builder.py:245-261)
        self.actor_mlp = mlp()
        self.critic_mlp = mlp()
        self.value = nn.Linear(units[-1], 1)
        self.mu = nn.Linear(units[-1], actions_num)
        for m in self.modules():                                   # :273-277 default init, zero biases
            if isinstance(m, nn.Linear):
                nn.init.zeros_(m.bias)
        self.sigma = nn.Parameter(torch.full((actions_num,), float(sigma_val)), requires_grad=False)   # amp_network_builder.py:22-27

    def eval_actor(self, obs):
        mu = self.mu(self.actor_mlp(obs))
        return mu, mu * 0.0 + self.sigma                           # amp_network_builder.py:142-148

    def eval_critic(self, obs):
        return self.value(self.critic_mlp(obs))

    def state_dict_ref(self):
        return {"a2c_network." + k: v.detach().clone() for k, v in self.state_dict().items()}

    @staticmethod
    def neglogp(x, mean, std, logstd):
        return 0.5 * (((x - mean) / std) ** 2).sum(dim=-1) + 0.5 * math.log(2.0 * math.pi) * x.size()[-1] + logstd.sum(dim=-1)

    def forward(self, d):
        """ModelA2CContinuousLogStd.Network.forward (rl_games 3P)."""
        mu, logstd = self.eval_actor(d["obs"])
        value = self.eval_critic(d["obs"])
        sigma = torch.exp(logstd)
        if d.get("is_train", True):
            entropy = (0.5 + 0.5 * math.log(2 * math.pi) + logstd).sum(dim=-1)
            return {"prev_neglogp": torch.squeeze(self.neglogp(d["prev_actions"], mu, sigma, logstd)), "values": value,
                    "entropy": entropy, "mus": mu, "sigmas": sigma}
        noise = d["noise"] if d.get("noise") is not None else torch.randn_like(mu)
        action = mu + sigma * noise                                 # Normal(mu, sigma).sample()
        return {"neglogpacs": torch.squeeze(self.neglogp(action, mu, sigma, logstd)), "values": value, "actions": action,
                "mus": mu, "sigmas": sigma}


def policy_kl(p0_mu, p0_sigma, p1_mu, p1_sigma, reduce=True):
    """rl_games torch_ext.policy_kl."""
    c1 = torch.log(p1_sigma / p0_sigma + 1e-5)
    c2 = (p0_sigma ** 2 + (p1_mu - p0_mu) ** 2) / (2.0 * (p1_sigma ** 2 + 1e-5))
    kl = (c1 + c2 + -0.5).sum(dim=-1)
    return kl.mean() if reduce else kl


def swap_and_flatten01(arr):
    s = arr.size()
    return arr.transpose(0, 1).reshape(s[0] * s[1], *s[2:])


class OracleEnv:
    """HumanoidIm over a RecordedRollout, on the CPU, with the reference's step order."""

    def __init__(self, rollout, reset_body_ids, track_body_ids):
        self.r = rollout
        self.num_envs = rollout.num_envs
        self.frame = 0
        self.rb = rollout.data["rb"][0].clone()
        self.progress_buf = rollout.init_progress.clone()
        self.reset_ids, self.track_ids = reset_body_ids, track_body_ids
        self.term_dist = torch.full((1, 24), 0.25)
        self.obs_buf = torch.zeros(self.num_envs, 934)
        self.dt = rollout.dt

    def _ref(self, d):
        return {k: v[self.frame] for k, v in d.items()}

    def reset(self, env_ids=None):
        if env_ids is None:
            env_ids = torch.arange(self.num_envs)
        if len(env_ids) > 0:
            self.rb[env_ids] = self.r.data["reset_rb"][self.frame][env_ids]
            self.progress_buf[env_ids] = 0
            bp, br, bv, ba = E.split_rb(self.rb[env_ids])
            rx = {k: v[env_ids] for k, v in self._ref(self.r.ref_next_reset).items()}
            so = E.self_obs_smpl_max(bp, br, bv, ba)
            tb = self.track_ids
            to = E.im_obs_v6(bp[:, 0], br[:, 0], bp[:, tb], br[:, tb], bv[:, tb], ba[:, tb], rx["pos"][:, tb], rx["rot"][:, tb],
                             rx["vel"][:, tb], rx["ang"][:, tb], 1)
            self.obs_buf[env_ids] = torch.cat([so, to], dim=-1)
        return self.obs_buf

    def step(self, actions):
        actions = torch.clamp(actions, -1.0, 1.0)                        # VecTaskPython.step
        self.frame = (self.frame + 1) % self.r.num_frames                 # physics: recorded
        self.rb = self.r.data["rb"][self.frame].clone()
        self.progress_buf += 1
        t = self.progress_buf * self.dt + self.r.motion_start_times
        pass_time = t >= self.r.motion_lengths
        out = E.post_physics(self.rb, self._ref(self.r.ref_now), self._ref(self.r.ref_next), self.r.data["dof_force"][self.frame],
                             self.r.data["dof_vel"][self.frame], self.progress_buf, pass_time, self.reset_ids, self.track_ids,
                             self.term_dist)
        self.obs_buf = out["obs"]
        return self.obs_buf, out["rew"], out["reset"], {"terminate": out["terminate"], "reward_raw": out["raw"]}


class OracleCommonAgent:
    """phc/learning/common_agent.py restated on the CPU (fp32, eager)."""

    def __init__(self, config, env, units, seed=0, noise=None):
        self.env = env
        self.num_actors = env.num_envs
        self.horizon_length = config["horizon_length"]
        self.gamma, self.tau = config["gamma"], config["tau"]
        self.e_clip, self.critic_coef = config["e_clip"], config["critic_coef"]
        self.bounds_loss_coef = config.get("bounds_loss_coef", None)
        self.clip_value = config["clip_value"]
        self.grad_norm, self.truncate_grads = config["grad_norm"], config["truncate_grads"]
        self.mini_epochs_num, self.minibatch_size = config["mini_epochs"], config["minibatch_size"]
        self.normalize_advantage = config["normalize_advantage"]
        self.mixed_precision = bool(config.get("mixed_precision", False))        # bf16 autocast around model forward + losses (:426)
        self.batch_size = self.horizon_length * self.num_actors
        self.last_lr = float(config["learning_rate"])
        self.actions_num, obs_dim = 69, 934
        self.model = OracleNet(obs_dim, self.actions_num, units, config["network"]["mlp"]["activation"],
                               config["network"]["space"]["continuous"]["sigma_init"]["val"])
        self.running_mean_std = OracleRunningMeanStd((obs_dim,))
        self.value_mean_std = OracleRunningMeanStd((1,))
        self.optimizer = torch.optim.Adam(self.model.parameters(), self.last_lr, eps=1e-08, weight_decay=0.0)   # :66
        g = torch.Generator()
        g.manual_seed(seed)
        self._perm_gen = g
        self._idx_buf = torch.randperm(self.batch_size, generator=g)                                             # amp_datasets.py:7
        self.noise = noise                      # optional (epochs, T, N, A) tensor shared with the device run
        self.epoch = 0
        t, n = self.horizon_length, self.num_actors
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype)
        self.tensor_dict = {"obses": z(t, n, obs_dim), "rewards": z(t, n, 1), "values": z(t, n, 1), "neglogpacs": z(t, n),
                            "dones": z(t, n, dtype=torch.uint8), "actions": z(t, n, 69), "mus": z(t, n, 69), "sigmas": z(t, n, 69)}
        self.tensor_dict["next_obses"] = torch.zeros_like(self.tensor_dict["obses"])                              # :92-98
        self.tensor_dict["next_values"] = torch.zeros_like(self.tensor_dict["values"])
        self.update_list = ["actions", "neglogpacs", "values", "mus", "sigmas"]
        self.tensor_list = self.update_list + ["obses", "dones", "next_obses"]
        self.obs = None
        self.grad_norms = []

    def set_eval(self):
        self.model.eval(); self.running_mean_std.eval(); self.value_mean_std.eval()

    def set_train(self):
        self.model.train(); self.running_mean_std.train(); self.value_mean_std.train()

    # :262-288
    def get_action_values(self, obs, noise):
        processed = self.running_mean_std(obs)
        self.model.eval()
        with torch.no_grad():
            res = self.model({"is_train": False, "prev_actions": None, "obs": processed, "noise": noise})
        res["values"] = self.value_mean_std(res["values"], True)
        return res

    # :551-562
    def _eval_critic(self, obs):
        self.model.eval()
        value = self.model.eval_critic(self.running_mean_std(obs))
        return self.value_mean_std(value, True)

    # :290-355
    def play_steps(self):
        self.set_eval()
        done_indices = []
        td = self.tensor_dict
        for n in range(self.horizon_length):
            self.obs = self.env.reset(done_indices)
            td["obses"][n, :] = self.obs
            noise = (self.noise(self.epoch, n) if callable(self.noise) else self.noise[self.epoch, n]) if self.noise is not None else None
            res = self.get_action_values(self.obs, noise)
            for k in self.update_list:
                td[k][n, :] = res[k]
            self.obs, rewards, self.dones, infos = self.env.step(res["actions"])
            rewards = rewards.unsqueeze(1)
            td["rewards"][n, :] = rewards
            td["next_obses"][n, :] = self.obs
            td["dones"][n, :] = self.dones
            terminated = infos["terminate"].float().unsqueeze(-1)
            next_vals = self._eval_critic(self.obs)
            next_vals *= (1.0 - terminated)
            td["next_values"][n, :] = next_vals
            all_done_indices = self.dones.nonzero(as_tuple=False)
            done_indices = all_done_indices[:, 0]
        mb_fdones = td["dones"].float()
        mb_advs = E.gae(mb_fdones, td["values"], td["rewards"], td["next_values"], self.gamma, self.tau)       # :493-505
        mb_returns = mb_advs + td["values"]
        batch_dict = {k: swap_and_flatten01(td[k]) for k in self.tensor_list}
        batch_dict["returns"] = swap_and_flatten01(mb_returns)
        batch_dict["advs_raw"] = swap_and_flatten01(mb_advs)
        return batch_dict

    # :357-398, :589-599
    def prepare_dataset(self, bd):
        advantages = torch.sum(bd["returns"] - bd["values"], axis=1)
        if self.normalize_advantage:
            advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
        values = self.value_mean_std(bd["values"])
        returns = self.value_mean_std(bd["returns"])
        self.values_dict = {"old_values": values, "old_logp_actions": bd["neglogpacs"], "advantages": advantages, "returns": returns,
                            "actions": bd["actions"], "obs": bd["obses"], "mu": bd["mus"], "sigma": bd["sigmas"]}
        return self.values_dict

    def _get_item(self, idx):                                                                                    # amp_datasets.py:81-100
        start, end = idx * self.minibatch_size, (idx + 1) * self.minibatch_size
        sample_idx = self._idx_buf[start:end]
        out = {k: v[sample_idx] for k, v in self.values_dict.items()}
        if end >= self.batch_size:
            self._idx_buf = torch.randperm(self.batch_size, generator=self._perm_gen)
        return out

    # :400-491, :512-520, :564-587
    def calc_gradients(self, d):
        self.set_train()
        obs_batch = self.running_mean_std(d["obs"])
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=self.mixed_precision):
            return self._calc_gradients_inner(d, obs_batch)

    def _calc_gradients_inner(self, d, obs_batch):
        res = self.model({"is_train": True, "prev_actions": d["actions"], "obs": obs_batch})
        action_log_probs, values, mu, sigma = res["prev_neglogp"], res["values"], res["mus"], res["sigmas"]
        ratio = torch.exp(d["old_logp_actions"] - action_log_probs)
        surr1 = d["advantages"] * ratio
        surr2 = d["advantages"] * torch.clamp(ratio, 1.0 - self.e_clip, 1.0 + self.e_clip)
        a_loss = torch.max(-surr1, -surr2)
        if self.clip_value:
            vpc = d["old_values"] + (values - d["old_values"]).clamp(-self.e_clip, self.e_clip)
            c_loss = torch.max((values - d["returns"]) ** 2, (vpc - d["returns"]) ** 2)
        else:
            c_loss = (d["returns"] - values) ** 2
        if self.bounds_loss_coef is not None:
            b_loss = (torch.clamp_max(mu + 1.0, 0.0) ** 2 + torch.clamp_min(mu - 1.0, 0.0) ** 2).sum(axis=-1)
        else:
            b_loss = torch.zeros(1)
        a_loss, c_loss, b_loss, entropy = torch.mean(a_loss), torch.mean(c_loss), torch.mean(b_loss), torch.mean(res["entropy"])
        loss = a_loss + self.critic_coef * c_loss - 0.0 * entropy + (self.bounds_loss_coef or 0.0) * b_loss
        for p in self.model.parameters():
            p.grad = None
        with torch.autocast("cpu", enabled=False):
            loss.backward()
        gn = None
        if self.truncate_grads:
            gn = nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_norm)
        self.optimizer.step()
        with torch.no_grad():
            kl = policy_kl(mu.detach(), sigma.detach(), d["mu"], d["sigma"], True)
        self.grad_norms.append(float(gn) if gn is not None else float("nan"))
        return {"actor_loss": a_loss.detach(), "critic_loss": c_loss.detach(), "b_loss": b_loss.detach(), "kl": kl, "grad_norm": gn}

    # :191-260
    def train_epoch(self, max_minibatches=None):
        if self.obs is None:
            self.obs = self.env.reset()
        t0 = time.time()
        with torch.no_grad():
            batch_dict = self.play_steps()
        t1 = time.time()
        self.set_train()
        self.prepare_dataset(batch_dict)
        infos = []
        nmb = self.batch_size // self.minibatch_size
        done = 0
        for _ in range(self.mini_epochs_num):
            for i in range(nmb):
                if max_minibatches is not None and done >= max_minibatches:
                    break
                infos.append(self.calc_gradients(self._get_item(i)))
                done += 1
        t2 = time.time()
        self.epoch += 1
        return {"batch_dict": batch_dict, "infos": infos, "play_time": t1 - t0, "update_time": t2 - t1, "minibatches": done}


class OracleNetZ(nn.Module):
    """AMPZBuilder.Network, z_type 'vae' with a learned prior, non-RNN branch
    (phc/learning/amp_network_z_builder.py:24-69, 79-121, 226-248, 326-339, 422-467, 469-580)."""

    def __init__(self, self_obs_size=358, task_obs_size=576, actions_num=69, units=(3096, 2048, 1024), task_units=(1536, 1024, 512),
                 embedding_size=32, var_clamp_max=2.0, sigma_val=-2.9):
        super().__init__()
        self.self_obs_size, self.task_obs_size, self.embedding_size = self_obs_size, task_obs_size, embedding_size
        self.var_clamp_max = var_clamp_max
        E = embedding_size

        def mlp(i, us):
            layers = []
            for u in us:
                layers += [nn.Linear(i, u), nn.SiLU()]
                i = u
            return nn.Sequential(*layers)
        self.actor_mlp = mlp(self_obs_size + E, units)
        self.critic_mlp = mlp(self_obs_size + E, units)
        self.value = nn.Linear(units[-1], 1)
        self.mu = nn.Linear(units[-1], actions_num)
        self.sigma = nn.Parameter(torch.full((actions_num,), float(sigma_val)), requires_grad=False)
        self.z_mlp = mlp(self_obs_size + task_obs_size, task_units)
        self.z_mlp.append(nn.Linear(task_units[-1], 5 * E))
        self.z_mu = nn.Linear(5 * E, E)
        self.z_logvar = nn.Linear(5 * E, E)
        self.z_prior = mlp(self_obs_size, task_units)
        self.z_prior_mu = nn.Linear(task_units[-1], E)
        self.z_prior_logvar = nn.Linear(task_units[-1], E)
        self.critic_z_mlp = mlp(self_obs_size + task_obs_size, task_units)
        self.critic_z_mlp.append(nn.Linear(task_units[-1], E))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.zeros_(m.bias)

    def state_dict_ref(self):
        return {"a2c_network." + k: v.detach().clone() for k, v in self.state_dict().items()}

    def form_embedding(self, task_out_z, noise):
        vae_mu = self.z_mu(task_out_z)
        vae_log_var = torch.clamp(self.z_logvar(task_out_z), min=-5, max=self.var_clamp_max)
        z = vae_mu + torch.exp(0.5 * vae_log_var) * noise
        return z, {"vae_mu": vae_mu, "vae_log_var": vae_log_var, "noise": noise}

    def eval_actor(self, obs, noise):
        self_obs = obs[:, :self.self_obs_size]
        z, extra = self.form_embedding(self.z_mlp(obs), noise)
        mu = self.mu(self.actor_mlp(torch.cat([self_obs, z], dim=-1)))
        return mu, mu * 0.0 + self.sigma, extra

    def compute_prior(self, obs):
        lat = self.z_prior(obs[:, :self.self_obs_size])
        return self.z_prior_mu(lat), torch.clamp(self.z_prior_logvar(lat), min=-5, max=self.var_clamp_max)

    def eval_critic(self, obs):
        self_obs = obs[:, :self.self_obs_size]
        return self.value(self.critic_mlp(torch.cat([self_obs, self.critic_z_mlp(obs)], dim=-1)))


class OracleNetSept(nn.Module):
    """AMPSeptBuilder.Network (phc/learning/amp_network_sept_builder.py:19-165), the terrain-task policy (learning/pulse_z_terrain.yaml): ONE
    task MLP (the constructor builds it twice into the same attribute, :33-36) over the task observation, shared by actor and critic,
    whose MLPs read cat(self_obs, task_out)."""

    def __init__(self, self_obs_size=358, task_obs_size=1044, actions_num=32, units=(2048, 1024, 512), task_units=(512, 256), sigma_val=-1.0):
        super().__init__()
        self.self_obs_size, self.task_obs_size = self_obs_size, task_obs_size

        def mlp(i, us):
            layers = []
            for u in us:
                layers += [nn.Linear(i, u), nn.SiLU()]
                i = u
            return nn.Sequential(*layers)
        cat = self_obs_size + task_units[-1]
        self.actor_mlp = mlp(cat, units)
        self.critic_mlp = mlp(cat, units)
        self.value = nn.Linear(units[-1], 1)
        self.mu = nn.Linear(units[-1], actions_num)
        self.sigma = nn.Parameter(torch.full((actions_num,), float(sigma_val)), requires_grad=False)
        self._task_mlp = mlp(task_obs_size, task_units)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.zeros_(m.bias)

    def state_dict_ref(self):
        return {"a2c_network." + k: v.detach().clone() for k, v in self.state_dict().items()}

    def _cat(self, obs):
        s, t = self.self_obs_size, self.task_obs_size
        assert obs.shape[-1] == s + t
        return torch.cat([obs[:, :s], self._task_mlp(obs[:, s:s + t])], dim=-1)

    def eval_actor(self, obs):
        mu = self.mu(self.actor_mlp(self._cat(obs)))
        return mu, mu * 0.0 + self.sigma

    def eval_critic(self, obs):
        return self.value(self.critic_mlp(self._cat(obs)))


def kl_multi(qm, qv, pm, pv):
    """phc/learning/loss_functions.py:3-10 (pinned: tests/golden/rms.npz kl_multi)."""
    return (0.5 * (pv - qv + qv.exp() / pv.exp() + (qm - pm).pow(2) / pv.exp() - 1)).sum(-1)


def oracle_optimize_kin(net, obs, gt_action, progress_buf, noise, horizon, kld_coefficient=0.01, ar1_coefficient=0.005,
                        use_ar1_prior=True, use_vae_prior_regu=False):
    """AMPAgent._optimize_kin, phc/learning/amp_agent.py:771-849 (z_type 'vae', use_vae_prior), up to and
    including kin_loss.backward().  ``obs`` is the already-normalised minibatch, rows ordered env-major
    (minibatch // horizon sequences of ``horizon`` steps).  Returns the info dict; gradients are left in net."""
    mb = obs.shape[0]
    pred_action, _, extra = net.eval_actor(obs, noise)
    kin_action_loss = torch.norm(pred_action - gt_action, dim=-1).mean()
    vae_mu, vae_log_var = extra["vae_mu"], extra["vae_log_var"]
    prior_mu, prior_log_var = net.compute_prior(obs)
    KLD = kl_multi(vae_mu, vae_log_var, prior_mu, prior_log_var).mean()
    ar1_prior, regu_prior = 0, 0
    info = {}
    if use_ar1_prior:
        time_zs = vae_mu.view(mb // horizon, horizon, -1)
        phi = 0.99
        error = time_zs[:, 1:] - time_zs[:, :-1] * phi
        idxes = progress_buf.view(mb // horizon, horizon, -1)
        not_consecs = ((idxes[:, 1:] - idxes[:, :-1]) != 1).view(-1)
        error = error.reshape(-1, error.shape[-1]).clone()
        error[not_consecs] = 0
        starteres = ((idxes <= 2)[:, 1:] + (idxes <= 2)[:, :-1]).view(-1)
        error[starteres] = 0
        ar1_prior = torch.norm(error, dim=-1).mean()
        info["kin_ar1"] = ar1_prior.detach()
    if use_vae_prior_regu:
        regu_prior = ((prior_mu ** 2).mean() + (vae_mu ** 2).mean()) * 0.001 + ((prior_log_var ** 2).mean() + (vae_log_var ** 2).mean()) * 0.001
    kin_loss = kin_action_loss + KLD * kld_coefficient + ar1_prior * ar1_coefficient + regu_prior * 0.005
    for p in net.parameters():
        p.grad = None
    kin_loss.backward()
    info.update({"kin_action_loss": kin_action_loss.detach(), "kin_KLD": KLD.detach(), "kin_loss": kin_loss.detach()})
    return info


def oracle_compute_z_actions(net, obs_buf, running_mean, running_var, action_z):
    """HumanoidZ.compute_z_actions, phc/env/tasks/humanoid_z.py:81-155 (z_type 'vae', use_vae_prior, not z_all),
    with the frozen sub-networks of network_loader.load_z_decoder (:139-176)."""
    with torch.no_grad():
        s = net.self_obs_size
        self_obs = (obs_buf[:, :s] - running_mean.float()[:s]) / torch.sqrt(running_var.float()[:s] + 1e-05)
        prior_mu = net.z_prior_mu(net.z_prior(self_obs))
        z = prior_mu + action_z
        self_obs = torch.clamp(self_obs, min=-5.0, max=5.0)
        return net.mu(net.actor_mlp(torch.cat([self_obs, z], dim=-1)))


class OracleDisc(nn.Module):
    """AMPBuilder.Network._build_disc / eval_disc, phc/learning/amp_network_builder.py:213-249."""

    def __init__(self, amp_dim, units=(1024, 512)):
        super().__init__()
        layers, i = [], amp_dim
        for u in units:
            layers += [nn.Linear(i, u), nn.ReLU()]
            i = u
        self._disc_mlp = nn.Sequential(*layers)
        self._disc_logits = nn.Linear(i, 1)
        for m in self._disc_mlp.modules():
            if isinstance(m, nn.Linear):
                nn.init.zeros_(m.bias)
        nn.init.uniform_(self._disc_logits.weight, -1.0, 1.0)
        nn.init.zeros_(self._disc_logits.bias)

    def eval_disc(self, amp_obs):
        return self._disc_logits(self._disc_mlp(amp_obs))

    def state_dict_ref(self):
        return {"a2c_network." + k: v.detach().clone() for k, v in self.state_dict().items()}


def oracle_disc_loss(disc, amp_obs, amp_obs_replay, amp_obs_demo, disc_logit_reg=0.01, disc_grad_penalty=5.0, disc_weight_decay=0.0001):
    """AMPAgent._disc_loss, phc/learning/amp_agent.py:895-952 (inputs already normalised)."""
    amp_obs_demo = amp_obs_demo.clone().requires_grad_(True)
    disc_agent_logit = torch.cat([disc.eval_disc(amp_obs), disc.eval_disc(amp_obs_replay)], dim=0)
    disc_demo_logit = disc.eval_disc(amp_obs_demo)
    bce = torch.nn.BCEWithLogitsLoss()
    disc_loss_agent = bce(disc_agent_logit, torch.zeros_like(disc_agent_logit))
    disc_loss_demo = bce(disc_demo_logit, torch.ones_like(disc_demo_logit))
    disc_loss = 0.5 * (disc_loss_agent + disc_loss_demo)
    logit_weights = torch.flatten(disc._disc_logits.weight)
    disc_logit_loss = torch.sum(torch.square(logit_weights))
    disc_loss = disc_loss + disc_logit_reg * disc_logit_loss
    disc_demo_grad = torch.autograd.grad(disc_demo_logit, amp_obs_demo, grad_outputs=torch.ones_like(disc_demo_logit), create_graph=True,
                                         retain_graph=True, only_inputs=True)[0]
    disc_grad_penalty_v = torch.mean(torch.sum(torch.square(disc_demo_grad), dim=-1))
    disc_loss = disc_loss + disc_grad_penalty * disc_grad_penalty_v
    if disc_weight_decay != 0:
        ws = [torch.flatten(m.weight) for m in disc._disc_mlp.modules() if isinstance(m, nn.Linear)] + [torch.flatten(disc._disc_logits.weight)]
        disc_loss = disc_loss + disc_weight_decay * torch.sum(torch.square(torch.cat(ws, dim=-1)))
    agent_acc = torch.mean((disc_agent_logit < 0).float())
    demo_acc = torch.mean((disc_demo_logit > 0).float())
    return {"disc_loss": disc_loss, "disc_grad_penalty": disc_grad_penalty_v.detach(), "disc_logit_loss": disc_logit_loss.detach(),
            "disc_agent_acc": agent_acc, "disc_demo_acc": demo_acc, "disc_agent_logit": disc_agent_logit.detach(),
            "disc_demo_logit": disc_demo_logit.detach()}


class OracleReplayBuffer:
    """phc/learning/replay_buffer.py:3-88 restated (single tensor per key, same RNG draws: torch.randperm on the CPU
    default generator at construction and whenever the sampling cursor wraps)."""

    def __init__(self, buffer_size, device="cpu"):
        self.size, self.device = buffer_size, device
        self.head, self.total, self.buf = 0, 0, None
        self.perm = torch.randperm(buffer_size)
        self.cursor = 0

    def store(self, data):
        if self.buf is None:
            self.buf = {k: torch.zeros((self.size,) + tuple(v.shape[1:]), device=self.device) for k, v in data.items()}
        n = next(iter(data.values())).shape[0]
        assert n <= self.size
        for k, dst in self.buf.items():
            first = min(n, self.size - self.head)
            dst[self.head:self.head + first] = data[k][:first]
            if n > first:
                dst[0:n - first] = data[k][first:]
        self.head = (self.head + n) % self.size
        self.total += n

    def sample(self, n):
        pos = torch.arange(self.cursor, self.cursor + n) % self.size
        rows = self.perm[pos]
        if self.total < self.size:
            rows = rows % self.head
        out = {k: v[rows] for k, v in self.buf.items()}
        self.cursor += n
        if self.cursor >= self.size:
            self.perm[:] = torch.randperm(self.size)
            self.cursor = 0
        return out


def oracle_disc_rewards(disc, amp_rms, amp_obs, disc_reward_scale=2.0):
    """AMPAgent._calc_disc_rewards, phc/learning/amp_agent.py:1027-1041 (norm_disc_reward False; normaliser in eval mode)."""
    was = amp_rms.training
    amp_rms.eval()
    with torch.no_grad():
        logits = disc.eval_disc(amp_rms(amp_obs))
        prob = 1 / (1 + torch.exp(-logits))
        r = -torch.log(torch.maximum(1 - prob, torch.tensor(0.0001))) * disc_reward_scale
    amp_rms.train(was)
    return r


class OracleAMPModel(nn.Module):
    """ModelAMPContinuous.Network (phc/learning/amp_models.py:16-43) over the restated policy / discriminator nets: the
    rl_games forward (OracleNet.forward) plus the three discriminator logits in training mode."""

    def __init__(self, net, disc):
        super().__init__()
        self.a2c_network = net
        self.disc = disc
        net.eval_disc = disc.eval_disc

    def forward(self, d):
        res = self.a2c_network(d)
        if d.get("is_train", True):
            res["disc_agent_logit"] = self.disc.eval_disc(d["amp_obs"])
            res["disc_agent_replay_logit"] = self.disc.eval_disc(d["amp_obs_replay"])
            res["disc_demo_logit"] = self.disc.eval_disc(d["amp_obs_demo"])
        return res


def oracle_amp_calc_gradients(model, optimizer, rms, rms_temp, amp_rms, d, cfg):
    """AMPAgent.calc_gradients, phc/learning/amp_agent.py:605-760 (PPO branch with the discriminator; fp32, single GPU):
    frozen-copy observation normalisation while the live statistics update (:594-601), the first amp_minibatch_size rows of the
    three AMP streams normalised in order (:621-629), loss = a + c_coef c - ent_coef H + b_coef b + disc_coef disc (:707-708),
    clip_grad_norm_ over ALL parameters, Adam.  ``cfg``: e_clip, critic_coef, entropy_coef, bounds_loss_coef, disc_coef,
    disc_logit_reg, disc_grad_penalty, disc_weight_decay, grad_norm, amp_minibatch_size, clip_value."""
    obs = d["obs"]
    obs_proc = rms_temp(obs)
    rms(obs)                                                    # running through the live mean/std, value unused
    b = cfg["amp_minibatch_size"]
    amp_obs, amp_replay, amp_demo = amp_rms(d["amp_obs"][0:b]), amp_rms(d["amp_obs_replay"][0:b]), amp_rms(d["amp_obs_demo"][0:b])
    amp_demo.requires_grad_(True)
    if cfg.get("autocast_bf16", False):           # amp_agent.py:671: autocast(enabled=self.mixed_precision) around forward + losses (bf16 here)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            return _amp_calc_gradients_tail(model, optimizer, d, cfg, obs_proc, amp_obs, amp_replay, amp_demo)
    return _amp_calc_gradients_tail(model, optimizer, d, cfg, obs_proc, amp_obs, amp_replay, amp_demo)


def _amp_calc_gradients_tail(model, optimizer, d, cfg, obs_proc, amp_obs, amp_replay, amp_demo):
    res = model({"is_train": True, "prev_actions": d["actions"], "obs": obs_proc, "amp_obs": amp_obs, "amp_obs_replay": amp_replay,
                 "amp_obs_demo": amp_demo})
    nlp, values, mu, sigma = res["prev_neglogp"], res["values"], res["mus"], res["sigmas"]
    ratio = torch.exp(d["old_logp_actions"] - nlp)
    surr1 = d["advantages"] * ratio
    surr2 = d["advantages"] * torch.clamp(ratio, 1.0 - cfg["e_clip"], 1.0 + cfg["e_clip"])
    a_loss = torch.max(-surr1, -surr2)
    if cfg.get("clip_value", False):
        vpc = d["old_values"] + (values - d["old_values"]).clamp(-cfg["e_clip"], cfg["e_clip"])
        c_loss = torch.max((values - d["returns"]) ** 2, (vpc - d["returns"]) ** 2)
    else:
        c_loss = (d["returns"] - values) ** 2
    b_loss = (torch.clamp_max(mu + 1.0, 0.0) ** 2 + torch.clamp_min(mu - 1.0, 0.0) ** 2).sum(axis=-1)
    a_loss, c_loss, b_loss, entropy = torch.mean(a_loss), torch.mean(c_loss), torch.mean(b_loss), torch.mean(res["entropy"])
    agent_cat = torch.cat([res["disc_agent_logit"], res["disc_agent_replay_logit"]], dim=0)
    demo_logit = res["disc_demo_logit"]
    bce = torch.nn.BCEWithLogitsLoss()
    disc_loss = 0.5 * (bce(agent_cat, torch.zeros_like(agent_cat)) + bce(demo_logit, torch.ones_like(demo_logit)))
    disc = model.disc
    logit_loss = torch.sum(torch.square(torch.flatten(disc._disc_logits.weight)))
    disc_loss = disc_loss + cfg["disc_logit_reg"] * logit_loss
    g = torch.autograd.grad(demo_logit, amp_demo, grad_outputs=torch.ones_like(demo_logit), create_graph=True, retain_graph=True, only_inputs=True)[0]
    penalty = torch.mean(torch.sum(torch.square(g), dim=-1))
    disc_loss = disc_loss + cfg["disc_grad_penalty"] * penalty
    if cfg["disc_weight_decay"] != 0:
        ws = [torch.flatten(m.weight) for m in disc._disc_mlp.modules() if isinstance(m, nn.Linear)] + [torch.flatten(disc._disc_logits.weight)]
        disc_loss = disc_loss + cfg["disc_weight_decay"] * torch.sum(torch.square(torch.cat(ws, dim=-1)))
    loss = a_loss + cfg["critic_coef"] * c_loss - cfg["entropy_coef"] * entropy + cfg["bounds_loss_coef"] * b_loss + cfg["disc_coef"] * disc_loss
    for p in model.parameters():
        p.grad = None
    with torch.autocast("cpu", enabled=False):
        loss.backward()
    with torch.no_grad():
        kl = policy_kl(mu.detach().float(), sigma.detach().float(), d["mu"], d["sigma"], True)
    gn = nn.utils.clip_grad_norm_(model.parameters(), cfg["grad_norm"])
    optimizer.step()
    return {"actor_loss": a_loss.detach(), "critic_loss": c_loss.detach(), "b_loss": b_loss.detach(), "entropy": entropy.detach(), "kl": kl,
            "disc_loss": disc_loss.detach(), "disc_grad_penalty": penalty.detach(), "grad_norm": gn}


def oracle_pnn_teacher_action(pnn_model, composer_model, num_prim, activation, obs, running_mean, running_var, has_lateral=False):
    """HumanoidImDistill.step's gt_action (phc/env/tasks/humanoid_im_distill.py:165-198, has_pnn branch, same obs settings for teacher
    and student) over load_pnn / load_mcp_mlp-shaped state dicts (phc/learning/network_loader.py:11-73; PNN forward
    phc/learning/pnn.py:84-131: without lateral links :125-131, with them :90-123 -- column c's second layer adds the bias-free
    ``u[c-1][j][0]`` images of every earlier column's first activation before its own activation; ``u[..][1]`` is unused, :104)."""
    act = {"relu": nn.ReLU, "silu": nn.SiLU}[activation]

    def seq(model, prefix, trailing_act):
        layers, i = [], 0
        while f"{prefix}.{2 * i}.weight" in model:
            w, b = model[f"{prefix}.{2 * i}.weight"], model[f"{prefix}.{2 * i}.bias"]
            lin = nn.Linear(w.shape[1], w.shape[0])
            with torch.no_grad():
                lin.weight.copy_(w); lin.bias.copy_(b)
            layers += [lin, act()]
            i += 1
        if not trailing_act:
            layers = layers[:-1]
        return nn.Sequential(*layers)

    with torch.no_grad():
        full_obs = torch.clamp((obs - running_mean.float()) / torch.sqrt(running_var.float() + 1e-05), min=-5.0, max=5.0)
        if has_lateral:
            cols, h1s = [], []
            for c in range(num_prim):
                net = seq(pnn_model, f"a2c_network.pnn.actors.{c}", False)
                assert len(net) == 5
                h1 = net[:2](full_obs)
                lat = [torch.nn.functional.linear(h1s[j], pnn_model[f"a2c_network.pnn.u.{c - 1}.{j}.0.weight"]) for j in range(len(h1s))]
                h2 = net[3](net[2](h1) + sum(lat))
                cols.append(net[4](h2))
                h1s.append(h1)
            x_all = torch.stack(cols, dim=1)
        else:
            x_all = torch.stack([seq(pnn_model, f"a2c_network.pnn.actors.{k}", False)(full_obs) for k in range(num_prim)], dim=1)
        weights = seq(composer_model, "a2c_network.composer", True)(full_obs)
        return torch.sum(weights[:, :, None] * x_all, dim=1)
