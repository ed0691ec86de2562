"""CPU, world_size 2 over gloo: the data-parallel protocol of pulse_amd/parallel.py (what runs over RCCL on the GPUs)."""
import os
import socket
import types

import torch
import torch.multiprocessing as mp

from pulse_amd.parallel import DistContext


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    ctx = DistContext(enabled=True, backend="gloo")
    assert (ctx.rank, ctx.world_size) == (rank, world)
    # setup_algo: rank 0's parameters / optimiser state win
    flat = torch.full((1000,), float(rank + 1))
    m = torch.full((1000,), float(rank))
    ctx.setup_algo(flat, (m,))
    ok = bool((flat == 1.0).all() and (m == 0.0).all())
    # envs are sharded: each rank has its own gradient, pre-scaled by 1/world in the slab reduce, SUM all-reduced
    g_local = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    g = g_local / world
    ctx.sync_gradients(g)
    expect = torch.arange(1000, dtype=torch.float32) * (1 + 2) / 2          # the average over ranks
    ok = ok and torch.allclose(g, expect)
    # scalar KL mean
    kl = ctx.average_value(torch.tensor(float(rank) + 0.5))
    ok = ok and abs(kl.item() - 1.0) < 1e-6
    # per-epoch stat sync: averaged buffers, summed frames
    st = types.SimpleNamespace(running_mean=torch.full((5,), float(rank), dtype=torch.float64),
                               running_var=torch.full((5,), 2.0 * rank + 1, dtype=torch.float64),
                               count=torch.tensor(10.0, dtype=torch.float64))
    frames = ctx.sync_stats([st, None], 131072)
    ok = ok and frames == 2 * 131072 and torch.allclose(st.running_mean, torch.full((5,), 0.5, dtype=torch.float64))
    ok = ok and torch.allclose(st.running_var, torch.full((5,), 2.0, dtype=torch.float64)) and st.count.item() == 10.0
    ok = ok and abs(ctx.max_over_ranks(1.0 + rank) - 2.0) < 1e-12
    ctx.barrier()
    ctx.shutdown()
    q.put((rank, ok))


def test_two_rank_protocol_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_disabled_context_is_identity():
    ctx = DistContext(enabled=False)
    t = torch.ones(3)
    assert ctx.sync_gradients(t) is t and ctx.average_value(t) is t and ctx.max_over_ranks(3.5) == 3.5
    assert ctx.sync_stats([], 7) == 7


def test_host_logic_without_gpu():
    """Host-side pieces that need no kernel: flatten order, dataset index logic, meters, config derivation."""
    from pulse_amd import configs
    from pulse_amd.learning import rlg
    t, n = 4, 3
    x = torch.arange(t * n * 2, dtype=torch.float32).reshape(n, t, 2)        # env-major physical storage
    tm = x.transpose(0, 1)                                                   # the reference's (T, N, .) view
    flat = rlg.swap_and_flatten01(tm)
    assert flat.data_ptr() == x.data_ptr()                                   # free reshape, no copy
    assert torch.equal(flat, x.reshape(n * t, 2))                            # row = env * T + t
    ds = rlg.AMPDataset(12, 4, False, False, "cpu", 4, generator=torch.Generator().manual_seed(3))
    ref = torch.randperm(12, generator=torch.Generator().manual_seed(3))
    ds.update_values_dict({"a": torch.arange(12)})
    assert len(ds) == 3
    got = torch.cat([ds[i]["idx"] for i in range(2)])
    assert torch.equal(got, ref[:8])
    assert torch.equal(ds.gather(2)["a"], ref[8:12])
    last = ds[2]["idx"]                                                       # consuming the last slice reshuffles (amp_datasets.py:91-92)
    assert torch.equal(last, ref[8:12]) and not torch.equal(ds._idx_buf, ref)
    m = rlg.AverageMeter((1,), 100, "cpu")
    vals = torch.tensor([[1.0], [2.0], [3.0], [4.0]])
    m.update_masked(vals, torch.tensor([True, False, True, False]))
    assert abs(m.get_mean().item() - 2.0) < 1e-6
    m.update_masked(vals, torch.tensor([False, False, False, False]))         # no finished episode: unchanged
    assert abs(m.get_mean().item() - 2.0) < 1e-6
    cfg, n_envs = configs.agent_config("cfg2")
    assert (n_envs, cfg["horizon_length"], cfg["minibatch_size"], cfg["mini_epochs"]) == (4096, 32, 16384, 6)
    assert cfg["network"]["mlp"]["units"] == [1024, 512] and cfg["e_clip"] == 0.2 and cfg["critic_coef"] == 5
