"""AMPAgent.train_epoch on a RECORDED environment trace, PyTorch CPU.

ORACLE / TEST INFRASTRUCTURE -- never imported by ``pulse_amd``.

The reference's ``im_amp`` agent (phc/learning/amp_agent.py) differs from CommonAgent in exactly the places this class
restates (line references into /root/reference/phc/learning/amp_agent.py):
  * :557-583  pre_epoch / post_epoch: ``running_mean_std_temp`` = frozen deep copy of the input normaliser;
  * :594-603  _preproc_obs(use_temp=True): the network is fed observations normalised by the frozen copy while the LIVE
              statistics keep being updated on every minibatch of every mini-epoch;
  * :341-439  play_steps: AMP observation windows recorded per step, bootstrap values zeroed on termination only;
  * :1011-1041 discriminator reward -log(max(1 - sigmoid(D(norm(x))), 1e-4)) * scale, mixed 0.5 / 0.5 with the task reward
              BEFORE GAE;
  * :474-484, :621-629  demo / replay rows ride in the dataset, the first ``amp_minibatch_size`` rows of each minibatch
              feed the discriminator, normalised agent -> replay -> demo (each pass updates the AMP statistics);
  * :605-760  calc_gradients = ``oracle_amp_calc_gradients`` (pinned bit-for-bit to the reference's method body in
              tests/test_oracle_vs_reference_learning.py).

What is recorded and what is recomputed: the env outputs of one rollout (observations, next observations, task rewards, dones,
terminate flags, AMP observation windows) are taken from the device run -- the env kernels are pinned against the env oracle by
their own tests, and the recorded physics ignores the action -- and EVERYTHING the agent computes from them is recomputed here
from the same initial weights, sampling noise, minibatch index lists and demo / replay rows.
"""
import copy

import torch

from . import agent_oracle as AO
from . import env_oracle as E


class OracleAMPAgent:
    def __init__(self, cfg, obs_dim, amp_dim, net_state, disc_state, units, disc_units, activation="relu"):
        self.cfg = cfg
        self.net = AO.OracleNet(obs_dim, 69, units, activation, cfg["network"]["space"]["continuous"]["sigma_init"]["val"])
        self.net.load_state_dict({k.replace("a2c_network.", ""): v.detach().cpu().clone() for k, v in net_state.items()})
        self.disc = AO.OracleDisc(amp_dim, units=disc_units)
        self.disc.load_state_dict({k.replace("a2c_network.", ""): v.detach().cpu().clone() for k, v in disc_state.items()})
        self.model = AO.OracleAMPModel(self.net, self.disc)
        self.running_mean_std = AO.OracleRunningMeanStd((obs_dim,))
        self.value_mean_std = AO.OracleRunningMeanStd((1,))
        self.amp_mean_std = AO.OracleRunningMeanStd((amp_dim,))
        self.optimizer = torch.optim.Adam(self.model.parameters(), float(cfg["learning_rate"]), eps=1e-08, weight_decay=0.0)
        self.grad_cfg = {"e_clip": cfg["e_clip"], "critic_coef": cfg["critic_coef"], "entropy_coef": cfg.get("entropy_coef", 0.0),
                         "bounds_loss_coef": cfg["bounds_loss_coef"], "disc_coef": cfg["disc_coef"], "disc_logit_reg": cfg["disc_logit_reg"],
                         "disc_grad_penalty": cfg["disc_grad_penalty"], "disc_weight_decay": cfg["disc_weight_decay"],
                         "grad_norm": cfg["grad_norm"], "amp_minibatch_size": cfg["amp_minibatch_size"], "clip_value": cfg["clip_value"],
                         "autocast_bf16": bool(cfg.get("mixed_precision", False))}

    def _mode(self, train):
        for m in (self.model, self.running_mean_std, self.value_mean_std, self.amp_mean_std):
            m.train(train)

    # :341-439 on recorded env outputs; ``rec`` tensors are time-major (T, N, .)
    def play_recorded(self, rec, noise):
        self._mode(False)
        T = rec["obses"].shape[0]
        out = {k: [] for k in ("actions", "neglogpacs", "values", "mus", "next_values")}
        with torch.no_grad():
            for n in range(T):
                res = self.net({"is_train": False, "obs": self.running_mean_std(rec["obses"][n]), "noise": noise[n]})
                res["values"] = self.value_mean_std(res["values"], True)
                for k in ("actions", "neglogpacs", "values", "mus"):
                    out[k].append(res[k])
                nv = self.value_mean_std(self.net.eval_critic(self.running_mean_std(rec["next_obses"][n])), True)
                out["next_values"].append(nv * (1.0 - rec["terminates"][n].float().unsqueeze(-1)))
            td = {k: torch.stack(v) for k, v in out.items()}
            amp = rec["amp_obs"]
            disc_r = AO.oracle_disc_rewards(self.disc, self.amp_mean_std, amp.reshape(-1, amp.shape[-1]),
                                            self.cfg["disc_reward_scale"]).reshape(T, -1, 1)
            mb_rewards = self.cfg["task_reward_w"] * rec["rewards"] + self.cfg["disc_reward_w"] * disc_r        # :1011-1016
            advs = E.gae(rec["dones"].float(), td["values"], mb_rewards, td["next_values"], self.cfg["gamma"], self.cfg["tau"])
        td.update({"disc_rewards": disc_r, "mb_rewards": mb_rewards, "advs": advs, "returns": advs + td["values"]})
        f = AO.swap_and_flatten01
        self.batch = {"obses": f(rec["obses"]), "actions": f(td["actions"]), "mus": f(td["mus"]), "neglogpacs": f(td["neglogpacs"]),
                      "values": f(td["values"]), "returns": f(td["returns"]), "amp_obs": f(amp)}
        return td

    # common_agent.py:357-398 + amp_agent.py:441-456
    def prepare_dataset(self):
        self._mode(True)
        b = self.batch
        adv = torch.sum(b["returns"] - b["values"], axis=1)
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        values = self.value_mean_std(b["values"])
        returns = self.value_mean_std(b["returns"])
        sig = (b["mus"] * 0.0 + self.net.sigma).exp()
        self.values_dict = {"old_values": values, "old_logp_actions": b["neglogpacs"], "advantages": adv, "returns": returns,
                            "actions": b["actions"], "obs": b["obses"], "mu": b["mus"], "sigma": sig, "amp_obs": b["amp_obs"]}
        return self.values_dict

    # :496-532 update loop over RECORDED minibatch index lists; demo / replay rows are given per dataset row
    def update(self, index_lists, amp_obs_demo_rows, amp_obs_replay_rows):
        self._mode(True)
        rms_temp = copy.deepcopy(self.running_mean_std)            # pre_epoch (:578-579); the rollout did not touch the statistics
        rms_temp.freeze()
        infos = []
        for idx in index_lists:
            d = {k: v[idx] for k, v in self.values_dict.items()}
            d["amp_obs_demo"] = amp_obs_demo_rows[idx]
            d["amp_obs_replay"] = amp_obs_replay_rows[idx]
            infos.append(AO.oracle_amp_calc_gradients(self.model, self.optimizer, self.running_mean_std, rms_temp, self.amp_mean_std, d,
                                                      self.grad_cfg))
        return infos
