"""Players: load a checkpoint and act, without a learner.

Mirrors rl_games' BasePlayer / phc/learning/common_player.py:35-175 (run loop) and phc/learning/amp_players.py:16-110
(AMPPlayerContinuous: restore also loads ``amp_input_mean_std``; discriminator reward for debugging, :112-160).
The networks are the same GEMM launch plans the agents use; the checkpoint layout is the reference's ({'model', 'running_mean_std',
'reward_mean_std', 'amp_input_mean_std'}, phc/learning/network_loader.py wire format).
"""
import torch

from .common_agent import CommonAgent
from .amp_agent import AMPAgent


class CommonPlayer:
    AGENT = CommonAgent

    def __init__(self, config):
        self.config = config
        # a player is an agent that never trains: construction builds the networks, normalisers and env plumbing once
        self._agent = self.AGENT("player", config)
        a = self._agent
        self.env, self.device = a.vec_env, a.ppo_device
        self.model, self.running_mean_std = a.model, a.running_mean_std
        self.actions_num, self.clip_actions = a.actions_num, a.clip_actions
        self.games_num = int(config.get("player", {}).get("games_num", 2000))
        self.is_determenistic = bool(config.get("player", {}).get("determenistic", True))
        self.max_steps = int(config.get("player", {}).get("max_steps", 108000))

    def restore(self, fn):
        ckpt = torch.load(fn, map_location=self.device, weights_only=False) if isinstance(fn, str) else fn
        self._agent.set_full_state_weights(ckpt)
        return ckpt

    def get_action(self, obs, is_determenistic=False):
        a = self._agent
        n = obs.shape[0]
        ws = a.model.workspace(n, train=False)
        a.set_eval()
        a._preproc_obs(obs, ws, n)
        a.model.forward(ws, n)
        mu = ws["mu"]
        act = mu.clone() if is_determenistic else mu + torch.exp(a.model.sigma) * torch.randn(n, self.actions_num, device=self.device)
        return torch.clamp(act, -1.0, 1.0) if self.clip_actions else act

    def _post_step(self, info):
        return

    def run(self, n_steps=None):
        """common_player.py:35-175 reduced to its data path: step all envs, reset the finished ones, average episode returns."""
        env = self.env
        obs = env.reset()
        n = env.num_envs
        cr = torch.zeros(n, device=self.device)
        sum_rewards, games = 0.0, 0
        for _ in range(int(n_steps or self.max_steps)):
            action = self.get_action(obs, self.is_determenistic)
            obs, r, done, info = env.step(action)
            cr += r
            self._post_step(info)
            d = done > 0
            k = int(d.sum())
            if k:
                sum_rewards += float(cr[d].sum())
                games += k
                cr.mul_(~d)
                obs = env.reset_masked(d)
            if games >= self.games_num:
                break
        return {"games": games, "av_reward": sum_rewards / max(1, games)}


class AMPPlayerContinuous(CommonPlayer):
    AGENT = AMPAgent

    def __init__(self, config):
        self._normalize_amp_input = config.get("normalize_amp_input", True)
        self._disc_reward_scale = config.get("disc_reward_scale", 2)
        super().__init__(config)

    def restore(self, fn):
        ckpt = super().restore(fn)                      # AMPAgent.set_full_state_weights also restores amp_input_mean_std (amp_players.py:64-73)
        return ckpt

    def _calc_disc_rewards(self, amp_obs):
        """amp_players.py:148-160 (debug read-out of the discriminator reward)."""
        a = self._agent
        if not getattr(a, "enable_disc", False):
            raise RuntimeError("this checkpoint / config has no discriminator")
        pad = torch.zeros(amp_obs.shape[0], a._amp_pitch, device=self.device)
        pad[:, :a._amp_dim] = amp_obs
        return a._calc_disc_rewards(pad)


class IMAMPPlayerContinuous(AMPPlayerContinuous):
    """phc/learning/im_amp_players.py: the player registered under ``im_amp`` -- evaluation over the motion library."""

    def __init__(self, config):
        from .im_amp import IMAmpAgent
        self.AGENT = IMAmpAgent
        super().__init__(config)

    def run(self, n_steps=None):
        return self._agent.eval(max_steps=n_steps)
