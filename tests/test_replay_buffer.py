"""CPU: the device ReplayBuffer (pulse_amd/learning/amp_agent.py) draws the same rows as the reference's
phc/learning/replay_buffer.py -- via the oracle restatement everywhere, and via the reference class itself when
/root/reference is mounted (this container only)."""
import importlib.util
import os

import pytest
import torch

from oracle import agent_oracle as AO
from pulse_amd.learning.amp_agent import ReplayBuffer

REF = "/root/reference/phc/learning/replay_buffer.py"


def _drive(make_ref, size, width, script, seed):
    torch.manual_seed(seed)
    ref = make_ref(size)
    torch.manual_seed(seed)
    mine = ReplayBuffer(size, width, "cpu")
    g = torch.Generator().manual_seed(seed + 1)
    for op, n in script:
        if op == "store":
            rows = torch.randn(n, width, generator=g)
            state = torch.get_rng_state()
            ref.store({"amp_obs": rows})
            torch.set_rng_state(state)
            mine.store(rows)
        else:
            state = torch.get_rng_state()
            a = ref.sample(n)["amp_obs"]
            after_ref = torch.get_rng_state()
            torch.set_rng_state(state)
            b = mine.sample(n)
            assert torch.equal(after_ref, torch.get_rng_state()), "RNG consumption differs"
            assert torch.equal(a, b), (op, n)
        assert ref.get_total_count() == mine.get_total_count() if hasattr(ref, "get_total_count") else ref.total == mine.get_total_count()


SCRIPT = [("store", 40), ("sample", 16), ("store", 70), ("sample", 64), ("sample", 50), ("store", 100), ("store", 37), ("sample", 100),
          ("sample", 99), ("store", 128), ("sample", 7), ("sample", 128), ("sample", 128)]


@pytest.mark.parametrize("seed", [0, 5])
def test_replay_buffer_matches_oracle(seed):
    _drive(lambda size: AO.OracleReplayBuffer(size), 128, 12, SCRIPT, seed)


@pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not mounted")
@pytest.mark.parametrize("seed", [0, 5])
def test_oracle_and_device_buffer_match_reference_class(seed):
    spec = importlib.util.spec_from_file_location("_ref_replay_buffer", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _drive(lambda size: mod.ReplayBuffer(size, "cpu"), 128, 12, SCRIPT, seed)
    # and the oracle itself against the reference class
    torch.manual_seed(seed)
    ref = mod.ReplayBuffer(64, "cpu")
    torch.manual_seed(seed)
    orc = AO.OracleReplayBuffer(64)
    g = torch.Generator().manual_seed(3)
    for n_store, n_sample in [(10, 5), (30, 40), (64, 64), (20, 63), (5, 2)]:
        rows = torch.randn(n_store, 6, generator=g)
        ref.store({"x": rows})
        orc.store({"x": rows})
        s = torch.get_rng_state()
        a = ref.sample(n_sample)["x"]
        torch.set_rng_state(s)
        b = orc.sample(n_sample)["x"]
        assert torch.equal(a, b)
