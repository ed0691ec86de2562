"""GPU-side generator of tests/golden/ckpt_pulse_small.pth (run on the MI355X box; uses pulse_amd only):

    python tools/make_ckpt_fixture.py gpurun_out/ckpt_pulse_small.pth

A small PULSE (amp_z, VAE latent 32) agent is trained for two epochs, its checkpoint (reference key layout: 'model',
'running_mean_std', 'reward_mean_std', 'epoch', 'frame') is saved WITHOUT the optimiser moments, and the frozen prior +
decoder of that checkpoint are evaluated inside HumanoidImZ.compute_z_actions on a few observations.  The CPU test
tests/test_checkpoint_format.py feeds the same file to the REFERENCE's loaders (phc/learning/network_loader.py) and the
reference's compute_z_actions source and compares the actions -- the checkpoint wire format and the decoder-in-env path
pinned against reference code in one go."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pulse_amd import configs
from pulse_amd.env.humanoid_z import HumanoidImZ
from pulse_amd.env.sim import RecordedMotion, RecordedRollout, RecordedSim

out = sys.argv[1]
dev = "cuda:0"
torch.manual_seed(123)
net = copy.deepcopy(configs.NETWORK_Z)
net["mlp"]["units"] = [96, 64, 48]
net["task_mlp"]["units"] = [80, 64, 32]
agent, _ = configs.make_agent("cfg3_small", device=dev, seed=77, network=net)
for e in range(2):
    agent.epoch_num = e + 1
    agent.train_epoch()
ck = agent.get_full_state_weights()
ck.pop("optimizer", None)
ck.pop("kin_optimizer", None)
to_cpu = lambda v: {k: to_cpu(x) for k, x in v.items()} if isinstance(v, dict) else (v.detach().cpu() if isinstance(v, torch.Tensor) else v)
ck = to_cpu(ck)
n = 24
rollout = RecordedRollout(n, 3, seed=5).to(dev)
sim = RecordedSim(rollout)
task = HumanoidImZ({"env": dict(configs.ENV_IM, embedding_size=32)}, sim, RecordedMotion(rollout, sim), device=dev)
task.initialize_z_models({k: (to_cpu(v) if isinstance(v, dict) else v) for k, v in ck.items()}, net)
task.reset()
az = 0.5 * torch.randn(n, 32)
act = task.compute_z_actions(az.to(dev)).clone()
ck["fixture"] = {"obs_buf": task.obs_buf.cpu().clone(), "action_z": az, "actions": act.cpu(), "network": net}
torch.save(ck, out)
print("saved", out, {k: (len(v) if isinstance(v, dict) else v) for k, v in ck.items() if k != "fixture"})
print(sorted(ck["model"].keys()))
