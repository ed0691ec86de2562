"""CPU: the oracle's self observation with shape / limb-weight rows reproduces the vectors the reference's own TorchScript functions wrote
(oracle/gen_golden.py: gen_env_shape_obs; phc/env/tasks/humanoid.py:1675-1731 with has_smpl_params / has_limb_weight_params, :1789-1849)."""
import os

import numpy as np
import torch

from oracle import env_oracle as E

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "env_shape_obs.npz"))
t = lambda k: torch.from_numpy(Z[k])


def test_shape_and_limb_weight_rows_bit_exact():
    rb = t("rb")
    bp, br, bv, ba = E.split_rb(rb)
    sh, lw, fs = t("smpl_params"), t("limb_weights"), t("force_sensor")
    for up, tag in ((True, ""), (False, "_noup")):
        for name, kw in (("both", dict(smpl_params=sh, limb_weight_params=lw)), ("shape", dict(smpl_params=sh)), ("limb", dict(limb_weight_params=lw))):
            got = E.self_obs_smpl_max_general(bp, br, bv, ba, True, True, up, **kw)
            assert torch.equal(got, t(f"self_obs_{name}{tag}")), (name, tag)
        got = E.self_obs_smpl_max_general(bp, br, bv, ba, True, True, up, force_sensor=fs, smpl_params=sh, limb_weight_params=lw)
        assert torch.equal(got, t(f"self_obs_v3_both{tag}"))
    assert t("self_obs_both").shape[1] == 358 + 11 + 10            # humanoid.py:653-661
