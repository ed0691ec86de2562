"""Driver for tools/pmc_env.sh: 12 launches of the fused env step (library mode, reward + reset + observations) at cfg2's width, and 12 of the
terrain step at 4096 envs x 1024 height points."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pulse_amd import configs
from pulse_amd._lib import PULSE_IM_RESET, PULSE_IM_REWARD, PULSE_IM_SELF_OBS, PULSE_IM_TASK_OBS, TASK_OBS, TASK_RESET, TASK_REWARD

dev = "cuda:0"
agent, _ = configs.make_agent("cfg2", device=dev, seed=1, reference="motion_lib")
agent.init_tensors()
agent.obs = agent.env_reset()
task = agent.vec_env.env.task
for _ in range(3):
    task.step(torch.zeros(task.num_envs, 69, device=dev))
full = PULSE_IM_REWARD | PULSE_IM_RESET | PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS
for _ in range(12):
    task._im_step(full, inc=0)
torch.cuda.synchronize()
agent_t, _ = configs.make_agent("terrain_z", device=dev, seed=1, num_envs_override=4096)
agent_t.init_tensors()
agent_t.obs = agent_t.env_reset()
tt = agent_t.vec_env.env.task
for _ in range(12):
    tt._task_step(TASK_OBS | TASK_REWARD | TASK_RESET)
torch.cuda.synchronize()
