"""Return-parity experiment (BASELINE.json north_star: "episode return within 1% of reference after 1000 PPO iterations").

    python tools/return_parity.py --iters 1000 --out profiles/r02_return_parity.json

Trains the HIP CommonAgent and the PyTorch-CPU oracle agent (oracle/agent_oracle.py) side by side at cfg1 scale (64 envs x horizon
16, [512, 512] actor / critic) on the action-dependent physics stand-in (pulse_pd_sim_step on the device, oracle/pd_sim_oracle.py on
the host): same motion library, same disturbance bank, same initial weights, same policy-sampling noise, same minibatch
permutations, episodes starting at motion time 0.  The two runs agree to fp32 round-off at first and then drift apart like any
two fp32 runs of a chaotic training loop; what is compared is the mean episode return over a window of epochs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


class EpisodeMeter:
    """Episode returns from the (T, N) reward / done tensors of consecutive rollouts (what game_rewards averages)."""

    def __init__(self, n):
        self.acc = torch.zeros(n, dtype=torch.float64)
        self.length = torch.zeros(n)
        self.finished = []            # (epoch, return, length)

    def feed(self, epoch, rewards, dones):
        r, d = rewards.reshape(rewards.shape[0], -1).double().cpu(), dones.reshape(dones.shape[0], -1).cpu().bool()
        for t in range(r.shape[0]):
            self.acc += r[t]
            self.length += 1
            for e in torch.nonzero(d[t]).flatten().tolist():
                self.finished.append((epoch, float(self.acc[e]), float(self.length[e])))
                self.acc[e] = 0
                self.length[e] = 0

    def mean_return(self, lo, hi):
        v = [r for e, r, _ in self.finished if lo <= e < hi]
        return (sum(v) / len(v), len(v)) if v else (float("nan"), 0)

    def mean_step_reward(self, lo, hi):
        v = [(r, l) for e, r, l in self.finished if lo <= e < hi]
        return sum(r for r, _ in v) / max(1.0, sum(l for _, l in v))


def run(iters, seed=7, device="cuda:0", window=None, log=print, side="both", envs=None):
    """side: 'both' trains the two agents in lockstep (short runs / tests); 'device' or 'oracle' trains one of them and returns its
    episode stream -- the CPU oracle needs no GPU, so the long experiment runs its two halves on different machines and
    ``merge`` compares them (seeds, noise and permutations are functions of (seed, epoch) only)."""
    from oracle import agent_oracle as AO
    from oracle import motion_oracle as MO
    from pulse_amd import configs
    window = window or max(1, iters // 5)
    cfg, num_envs = configs.agent_config("cfg1")
    if envs:
        num_envs = int(envs)
    T = cfg["horizon_length"]
    env_over = {"physics": "pd", "stateInit": "Start"}
    torch.set_num_threads(min(8, torch.get_num_threads()))       # the oracle's ops are tiny: a 128-thread pool only adds overhead
    agent = None
    if side in ("both", "device"):
        agent, _ = configs.make_agent("cfg1", device=device, seed=seed, reference="motion_lib", env_overrides=env_over, permutation_device="cpu",
                                      num_envs_override=num_envs)
    torch.manual_seed(seed)
    oenv = MO.make_agent_env(num_envs, T, seed, physics="pd", state_init_start=True)

    def noise(epoch, step=None):
        g = torch.Generator().manual_seed(seed * 100003 + epoch)
        z = torch.randn(T, num_envs, 69, generator=g)
        return z if step is None else z[step]
    cache = {}

    def noise_cached(epoch, step):
        if cache.get("e") != epoch:
            cache["e"], cache["z"], cache["zd"] = epoch, noise(epoch), None
        return cache["z"][step]
    oracle = AO.OracleCommonAgent(cfg, oenv, cfg["network"]["mlp"]["units"], seed=seed, noise=noise_cached)
    if agent is not None:
        agent.model.load_state_dict(oracle.model.state_dict_ref())        # same initial weights (CPU draw, seed-determined)

        def dev_noise(e, s):
            if cache.get("e") != e or cache.get("zd") is None:
                cache["e"], cache["z"] = e, noise(e)
                cache["zd"] = cache["z"].to(device)
            return cache["zd"][s]
        agent.noise_provider = dev_noise
    m_dev, m_ref = EpisodeMeter(num_envs), EpisodeMeter(num_envs)
    t_dev = t_ref = 0.0
    first_diff = None
    for it in range(iters):
        t0 = time.time()
        if agent is not None:
            agent.train_epoch()
            torch.cuda.synchronize()
            td = agent.experience_buffer.tensor_dict
            m_dev.feed(it, td["rewards"], td["dones"])
        t1 = time.time()
        if side in ("both", "oracle"):
            oracle.train_epoch()
            rd = oracle.tensor_dict
            m_ref.feed(it, rd["rewards"], rd["dones"])
        t2 = time.time()
        t_dev, t_ref = t_dev + (t1 - t0), t_ref + (t2 - t1)
        if side == "both" and first_diff is None and not torch.equal(td["dones"].cpu(), rd["dones"]):
            first_diff = it
        if it % max(1, iters // 20) == 0 or it == iters - 1:
            a, na = m_dev.mean_return(max(0, it - window), it + 1)
            b, nb = m_ref.mean_return(max(0, it - window), it + 1)
            log(f"[return_parity] iter {it}: device {a:.4f} ({na} eps)  oracle {b:.4f} ({nb} eps)  ({t_dev:.0f} s device, {t_ref:.0f} s oracle)")
    if side != "both":
        m = m_dev if side == "device" else m_ref
        from pulse_amd import kernels as K_
        return {"side": side, "iterations": iters, "seed": seed, "envs": num_envs, "finished": m.finished, "seconds": t_dev if side == "device" else t_ref,
                "torch": torch.__version__, "gemm_f32_arithmetic": K_.F32_MODE if side == "device" else "torch CPU fp32"}
    lo = iters - window
    a, na = m_dev.mean_return(lo, iters)
    b, nb = m_ref.mean_return(lo, iters)
    a0, _ = m_dev.mean_return(0, window)
    b0, _ = m_ref.mean_return(0, window)
    sa, sb = m_dev.mean_step_reward(lo, iters), m_ref.mean_step_reward(lo, iters)
    return {"iterations": iters, "config": "cfg1 (64 envs x horizon 16, [512, 512]) on the PD physics stand-in, motion-library reference",
            "window_epochs": window, "device_mean_episode_return": a, "oracle_mean_episode_return": b, "episodes_in_window": [na, nb],
            "relative_difference": abs(a - b) / abs(b), "device_mean_step_reward": sa, "oracle_mean_step_reward": sb,
            "relative_difference_step_reward": abs(sa - sb) / abs(sb),
            "first_window": {"device": a0, "oracle": b0}, "first_epoch_with_different_dones": first_diff,
            "seconds": {"device": t_dev, "oracle_cpu": t_ref}, "seed": seed}


def merge(dev_path, ref_path, window=None):
    """Compare the episode streams of a 'device' run and an 'oracle' run of the same (seed, iterations)."""
    d, r = json.load(open(dev_path)), json.load(open(ref_path))
    assert d["iterations"] == r["iterations"] and d["seed"] == r["seed"] and d.get("envs", 64) == r.get("envs", 64)
    iters = d["iterations"]
    window = window or max(1, iters // 5)
    md, mr = EpisodeMeter(1), EpisodeMeter(1)
    md.finished, mr.finished = [tuple(x) for x in d["finished"]], [tuple(x) for x in r["finished"]]
    lo = iters - window
    a, na = md.mean_return(lo, iters)
    b, nb = mr.mean_return(lo, iters)
    sa, sb = md.mean_step_reward(lo, iters), mr.mean_step_reward(lo, iters)
    curve = []
    for k in range(0, iters, max(1, iters // 10)):
        curve.append({"epochs": [k, min(iters, k + max(1, iters // 10))], "device": md.mean_return(k, k + max(1, iters // 10))[0],
                      "oracle": mr.mean_return(k, k + max(1, iters // 10))[0]})
    return {"iterations": iters, "config": f"cfg1 ({d.get('envs', 64)} envs x horizon 16, [512, 512]) on the PD physics stand-in, motion-library reference",
            "gemm_f32_arithmetic": d.get("gemm_f32_arithmetic", "unrecorded"),
            "window_epochs": window, "device_mean_episode_return": a, "oracle_mean_episode_return": b, "episodes_in_window": [na, nb],
            "relative_difference": abs(a - b) / abs(b), "device_mean_step_reward": sa, "oracle_mean_step_reward": sb,
            "relative_difference_step_reward": abs(sa - sb) / abs(sb), "learning_curve_mean_episode_return": curve,
            "first_window": {"device": md.mean_return(0, window)[0], "oracle": mr.mean_return(0, window)[0]},
            "seconds": {"device": d["seconds"], "oracle_cpu": r["seconds"]}, "seed": d["seed"],
            "where": {"device": "MI355X (gpurun)", "oracle": f"build container CPU, torch {r['torch']}"}}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--side", default="both", choices=["both", "device", "oracle"])
    ap.add_argument("--merge", nargs=2, default=None, metavar=("DEVICE_JSON", "ORACLE_JSON"))
    ap.add_argument("--out", default=None)
    ap.add_argument("--envs", type=int, default=None, help="number of envs (default: cfg1's 64)")
    a = ap.parse_args()
    if a.merge:
        res = merge(*a.merge)
    else:
        res = run(a.iters, seed=a.seed, side=a.side, log=lambda m: print(m, flush=True), envs=a.envs)
    print(json.dumps({k: v for k, v in res.items() if k != "finished"}, indent=1), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
