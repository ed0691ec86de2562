"""CPU twin of the action-dependent physics stand-in (pulse_pd_sim_step / pulse_amd/env/sim.py:PdSim).

ORACLE / TEST INFRASTRUCTURE -- never imported by ``pulse_amd``.  Same arithmetic, operation for operation, as the HIP kernel
(include/pulse_hip.h: pulse_pd_sim_args): per-joint error states under PD control towards ``sag + action_scale * action`` with a
recorded disturbance, semi-implicit Euler over ``substeps``; bodies are the tracked reference displaced by those errors.  There is
no reference counterpart (Isaac Gym's physics is out of scope): the twin exists so that a whole training run of the
HIP agent can be compared with the CPU oracle agent on a task whose return depends on the policy (tools/return_parity.py).
"""
import torch

from pulse_amd import synthetic as syn

from . import rotations as R
from .motion_oracle import OracleMotionEnv


class OraclePdMotionEnv(OracleMotionEnv):
    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.err = torch.zeros(self.n, syn.NUM_DOF)
        self.err_vel = torch.zeros(self.n, syn.NUM_DOF)
        self.sag, self.lever_dir = syn.pd_sim_tables()
        self.actions = torch.zeros(self.n, syn.NUM_DOF)

    def reset(self, env_ids, start_times):
        if len(env_ids) > 0:
            self.err[env_ids] = 0
            self.err_vel[env_ids] = 0
        return super().reset(env_ids, start_times)

    def _physics(self, tgt, f):
        c = syn.PD_SIM
        h = self.dt / float(c["substeps"])
        e, ev = self.err, self.err_vel
        te = self.sag + c["action_scale"] * self.actions
        for _ in range(c["substeps"]):
            acc = c["kp"] * (te - e) - c["kd"] * ev + self.bank["acc"][f]
            ev = ev + h * acc
            e = e + h * ev
        tq = c["kp"] * (te - e) - c["kd"] * ev
        self.err, self.err_vel = e, ev
        ref = self._records(tgt)
        n, j = self.n, ref.shape[1]
        e3, v3 = e.view(n, j - 1, 3), ev.view(n, j - 1, 3)
        u = self.lever_dir[1:].unsqueeze(0).expand(n, -1, -1)
        rb = ref.clone()
        rb[:, 1:, 0:3] = ref[:, 1:, 0:3] + c["lever"] * torch.cross(e3, u, dim=-1)
        rb[:, 1:, 3:7] = R.qmul(R.exp_map_to_q(e3.reshape(-1, 3)), ref[:, 1:, 3:7].reshape(-1, 4)).view(n, j - 1, 4)
        rb[:, 1:, 7:10] = ref[:, 1:, 7:10] + c["lever"] * torch.cross(v3, u, dim=-1)
        rb[:, 1:, 10:13] = ref[:, 1:, 10:13] + v3
        self.rb = rb
        self.dof_pos, self.dof_vel = tgt["dof_pos"] + e, tgt["dof_vel"] + ev
        return rb, tq
