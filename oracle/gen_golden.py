"""Generate tests/golden/*.npz from the REAL reference functions.

ORACLE / TEST INFRASTRUCTURE.  Runs only in the build container (needs
/root/reference, read-only).  The reference's TorchScript functions are loaded
by oracle/refload.py (unmodified text, executed eagerly on PyTorch CPU, fp32)
and evaluated on seeded synthetic inputs from pulse_amd/synthetic.py; inputs and
outputs are stored side by side so the fixtures are self-contained.

    python -m oracle.gen_golden            # rewrites tests/golden/*.npz

Fixtures (all small, < 1 MB each):
  rotations.npz   every quaternion / exp-map / 6-D op on the path + edge cases
  env_im.npz      self-obs, task-obs v6 / v7, imitation reward (+power), im-reset
  env_amp.npz     AMP per-frame observation (full 23 joints, 19-joint dof subset, global root)
  agent_math.npz  GAE discount_values, PPO actor/critic/bound losses, _calc_advs
  rms.npz         RunningMeanStd forward sequences (fp64 state), kl_multi
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import refload  # noqa: E402
from pulse_amd import synthetic as syn  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def _np(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        else:
            out[k] = np.asarray(v)
    return out


def gen_rotations():
    tu = refload.torch_utils()
    g = syn.make_generator(777)
    m = 257
    q = torch.randn(m, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    p = torch.randn(m, 4, generator=g)
    p = p / p.norm(dim=-1, keepdim=True)
    v = torch.randn(m, 3, generator=g)
    e = torch.randn(m, 3, generator=g) * torch.rand(m, 1, generator=g) * 3.0
    t = torch.rand(m, 1, generator=g)
    # edge cases
    q[0] = torch.tensor([0.0, 0.0, 0.0, 1.0])            # identity: sin_theta == 0
    q[1] = torch.tensor([0.0, 0.0, 0.0, -1.0])           # w = -1
    q[2] = torch.tensor([1.0, 0.0, 0.0, 0.0])            # pi rotation
    q[3] = torch.tensor([0.0, 0.0, 1e-6, 1.0])           # below the 1e-5 mask
    q[3] = q[3] / q[3].norm()
    q[4] = torch.tensor([0.0, 0.0, 0.0, 1.0000001])      # |w| > 1 -> NaN inside, masked
    p[5] = q[5]                                          # slerp of identical quats
    p[6] = -q[6]                                         # antipodal
    p[7] = q[7] + 1e-4 * torch.randn(4, generator=g)     # nearly identical -> lerp fallback
    p[7] = p[7] / p[7].norm()
    e[0] = 0.0                                           # zero exp-map -> 0/0 masked
    e[1] = torch.tensor([0.0, 0.0, 1e-6])
    e[2] = torch.tensor([0.0, 3.5, 0.0])                 # angle > pi wraps
    ang, ax = tu.quat_to_angle_axis(q)
    out = {
        "q": q, "p": p, "v": v, "e": e, "t": t,
        "quat_mul": tu.quat_mul(q, p),
        "quat_conjugate": tu.quat_conjugate(q),
        "my_quat_rotate": tu.my_quat_rotate(q, v),
        "quat_to_angle": ang, "quat_to_axis": ax,
        "quat_to_exp_map": tu.quat_to_exp_map(q),
        "quat_to_tan_norm": tu.quat_to_tan_norm(q),
        "exp_map_to_quat": tu.exp_map_to_quat(e),
        "slerp": tu.slerp(q, p, t),
        "calc_heading": tu.calc_heading(q),
        "calc_heading_quat": tu.calc_heading_quat(q),
        "calc_heading_quat_inv": tu.calc_heading_quat_inv(q),
        "normalize_angle": tu.normalize_angle(e[:, 0] * 3.0),
    }
    # secondary statement of the isaacgym algebra: poselib (16-multiply product)
    r3d = refload.importable_modules()["rotation3d"]
    out["poselib_quat_mul"] = r3d.quat_mul(q, p)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "rotations.npz"), **_np(out))


def gen_env_im(n=67):
    fn = refload.env_functions()
    g = syn.make_generator(1234)
    d = syn.env_step_inputs(g, n)
    rb = d["rb"]
    bp, br, bv, ba = rb[..., 0:3], rb[..., 3:7], rb[..., 7:10], rb[..., 10:13]
    rn, rx = d["ref_now"], d["ref_next"]
    empty = torch.zeros(n, 0)
    self_obs = fn["compute_humanoid_observations_smpl_max"](bp, br, bv, ba, empty, empty, True, True, True, False, False)
    self_obs_global_root = fn["compute_humanoid_observations_smpl_max"](bp, br, bv, ba, empty, empty, False, True, True, False, False)
    task_v6 = fn["compute_imitation_observations_v6"](bp[:, 0], br[:, 0], bp, br, bv, ba,
                                                      rx["pos"], rx["rot"], rx["vel"], rx["ang"], 1, True)
    tb = syn.VR_TRACK_BODY_IDS
    task_v7 = fn["compute_imitation_observations_v7"](bp[:, 0], br[:, 0], bp[:, tb], bv[:, tb],
                                                      rx["pos"][:, tb], rx["vel"][:, tb], 1, True)
    task_v6_vr = fn["compute_imitation_observations_v6"](bp[:, 0], br[:, 0], bp[:, tb], br[:, tb], bv[:, tb], ba[:, tb],
                                                         rx["pos"][:, tb], rx["rot"][:, tb], rx["vel"][:, tb], rx["ang"][:, tb], 1, True)
    specs = {"k_pos": 100.0, "k_rot": 10.0, "k_vel": 0.1, "k_ang_vel": 0.1,
             "w_pos": 0.5, "w_rot": 0.3, "w_vel": 0.1, "w_ang_vel": 0.1}
    rew, raw = fn["compute_imitation_reward"](bp[:, 0], br[:, 0], bp, br, bv, ba,
                                              rn["pos"], rn["rot"], rn["vel"], rn["ang"], specs)
    # power term exactly as HumanoidIm._compute_reward does it (humanoid_im.py:908-917)
    power = torch.abs(torch.multiply(d["dof_force"], d["dof_vel"])).sum(dim=-1)
    power_reward = -0.0005 * power
    power_reward[d["progress"] <= 3] = 0
    rew_total = rew + power_reward
    raw_total = torch.cat([raw, power_reward[:, None]], dim=-1)
    # reset exactly as HumanoidIm._compute_reset's else-branch (humanoid_im.py:1176-1186)
    rid = syn.RESET_BODY_IDS
    term_dist = torch.full((1, 24), 0.25)
    reset_buf = torch.zeros(n, dtype=torch.int64)
    contact = torch.zeros(n, 24, 3)
    reset, terminate = fn["compute_humanoid_im_reset"](reset_buf, d["progress"], contact, torch.tensor([7, 3, 8, 4]),
                                                       bp[:, rid].clone(), rn["pos"][:, rid].clone(), d["pass_time"],
                                                       True, term_dist[..., rid], False, False)
    reset_mean, terminate_mean = fn["compute_humanoid_im_reset"](reset_buf, d["progress"], contact, torch.tensor([7, 3, 8, 4]),
                                                                 bp[:, rid].clone(), rn["pos"][:, rid].clone(), d["pass_time"],
                                                                 True, term_dist[..., rid], False, True)
    out = {
        "rb": rb, "dof_force": d["dof_force"], "dof_vel": d["dof_vel"], "progress": d["progress"],
        "pass_time": d["pass_time"],
        "ref_now_pos": rn["pos"], "ref_now_rot": rn["rot"], "ref_now_vel": rn["vel"], "ref_now_ang": rn["ang"],
        "ref_next_pos": rx["pos"], "ref_next_rot": rx["rot"], "ref_next_vel": rx["vel"], "ref_next_ang": rx["ang"],
        "self_obs": self_obs, "self_obs_global_root": self_obs_global_root,
        "task_obs_v6": task_v6, "task_obs_v7_vr": task_v7, "task_obs_v6_vr": task_v6_vr,
        "reward_im": rew, "reward_raw_im": raw, "reward": rew_total, "reward_raw": raw_total,
        "reset": reset, "terminate": terminate, "reset_mean": reset_mean, "terminate_mean": terminate_mean,
    }
    np.savez_compressed(os.path.join(GOLDEN_DIR, "env_im.npz"), **_np(out))


def gen_env_variants(n=19):
    """The remaining observation variants: obs_v 1 / 2 / 3 / 8 / 9 (humanoid_im.py:1222-1325,1415-1540),
    self_obs_v 2 / 3 (humanoid.py:1734-1849), remove_base_rot (:1616-1620) and the non-upright forms of the shipped variants."""
    fn = refload.env_functions()
    g = syn.make_generator(777)
    T = 3
    rb = syn.rigid_body_state(g, n)
    bp, br, bv, ba = rb[..., 0:3].clone(), rb[..., 3:7].clone(), rb[..., 7:10].clone(), rb[..., 10:13].clone()
    ref = syn.rigid_body_state(g, n * T)
    rp, rr, rv, ra = ref[..., 0:3].clone(), ref[..., 3:7].clone(), ref[..., 7:10].clone(), ref[..., 10:13].clone()
    rp = rp.view(n, T, 24, 3) * 0.1 + bp.view(n, 1, 24, 3)                       # references near the simulated bodies
    rp = rp.reshape(n * T, 24, 3)
    tb = syn.VR_TRACK_BODY_IDS
    dof_pos = torch.randn(n, 69, generator=g) * 0.5
    ref_dof = torch.randn(n, 69, generator=g) * 0.5
    sub = lambda x, ids: x[:, ids].contiguous()
    one = lambda x: x.view(n, T, *x.shape[1:])[:, 0].contiguous()               # first future sample only (T = 1 inputs)
    out = {"rb": rb, "ref_pos": rp, "ref_rot": rr, "ref_vel": rv, "ref_ang": ra, "dof_pos": dof_pos, "ref_dof_pos": ref_dof,
           "T": torch.tensor(T)}
    out["remove_base_rot"] = fn["remove_base_rot"](br[:, 0].contiguous())
    for up in (True, False):
        tag = "" if up else "_noup"
        args = lambda ids, t: (bp[:, 0], br[:, 0], sub(bp, ids), sub(br, ids), sub(bv, ids), sub(ba, ids))
        full = list(range(24))
        for ver, name in ((1, "compute_imitation_observations"), (3, "compute_imitation_observations_v3"),
                          (6, "compute_imitation_observations_v6"), (9, "compute_imitation_observations_v9")):
            for ids, itag in ((full, ""), (tb, "_vr")):
                r4 = (sub(rp, ids), sub(rr, ids), sub(rv, ids), sub(ra, ids))
                if ver == 9:
                    r4 = (r4[0], r4[1], r4[2][:, 0].contiguous(), r4[3][:, 0].contiguous())
                out[f"v{ver}_T{T}{itag}{tag}"] = fn[name](*args(ids, T), *r4, T, up)
        r1 = (one(rp), one(rr), one(rv), one(ra))
        out[f"v8_T1{tag}"] = fn["compute_imitation_observations_v8"](*args(full, 1), *r1, 1, up)
        out[f"v7_T1_vr{tag}"] = fn["compute_imitation_observations_v7"](bp[:, 0], br[:, 0], sub(bp, tb), sub(bv, tb), sub(r1[0], tb), sub(r1[2], tb), 1, up)
        # v2 as HumanoidIm calls it (:755-758): dof subsets of the tracked bodies without the root
        ids = full
        dsel = lambda d: d.reshape(-1, 23, 3)[:, [i - 1 for i in ids[1:]], :].contiguous()
        out[f"v2_T1{tag}"] = fn["compute_imitation_observations_v2"](*args(ids, 1), dsel(dof_pos), *r1, dsel(ref_dof), 1, up)
        empty = torch.zeros(n, 0)
        for lro in (True, False):
            ltag = "" if lro else "_globalroot"
            out[f"self_obs{ltag}{tag}"] = fn["compute_humanoid_observations_smpl_max"](bp, br, bv, ba, empty, empty, lro, True, up, False, False)
            fs = torch.randn(n, 12, generator=torch.Generator().manual_seed(5))
            out["force_sensor"] = fs
            out[f"self_obs_v3{ltag}{tag}"] = fn["compute_humanoid_observations_smpl_max_v3"](bp, br, bv, ba, fs, empty, empty, lro, True, up, False, False)
            hist = syn.rigid_body_state(torch.Generator().manual_seed(9), n * T).view(n, T, 24, 13)
            out["rb_hist"] = hist
            if lro:          # with local_root_obs False the reference assigns a (B*T, 6) block into a (B, T, 6) slot and raises (:1766-1768)
                out[f"self_obs_v2{tag}"] = fn["compute_humanoid_observations_smpl_max_v2"](
                    hist[..., 0:3].contiguous(), hist[..., 3:7].contiguous(), hist[..., 7:10].contiguous(), hist[..., 10:13].contiguous(),
                    empty, empty, lro, True, up, False, False, T)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "env_variants.npz"), **_np(out))


def gen_env_shape_obs(n=19):
    """Shape / limb-weight rows of the self observation (has_smpl_params / has_limb_weight_params, humanoid.py:1724-1728, 1843-1847;
    enabled by robot has_shape_obs / has_weight_obs in the phc_shape_* configs)."""
    fn = refload.env_functions()
    g = syn.make_generator(4242)
    rb = syn.rigid_body_state(g, n)
    bp, br, bv, ba = rb[..., 0:3].clone(), rb[..., 3:7].clone(), rb[..., 7:10].clone(), rb[..., 10:13].clone()
    shapes = torch.randn(n, 11, generator=g)
    limbs = torch.rand(n, 10, generator=g) + 0.5
    fs = torch.randn(n, 12, generator=g)
    out = {"rb": rb, "smpl_params": shapes, "limb_weights": limbs, "force_sensor": fs}
    for up in (True, False):
        tag = "" if up else "_noup"
        for hs, hl, name in ((True, True, "both"), (True, False, "shape"), (False, True, "limb")):
            out[f"self_obs_{name}{tag}"] = fn["compute_humanoid_observations_smpl_max"](bp, br, bv, ba, shapes, limbs, True, True, up, hs, hl)
        out[f"self_obs_v3_both{tag}"] = fn["compute_humanoid_observations_smpl_max_v3"](bp, br, bv, ba, fs, shapes, limbs, True, True, up, True, True)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "env_shape_obs.npz"), **_np(out))


def gen_env_amp(n=67):
    """AMP per-frame observation from the reference's build_amp_observations_smpl (+ dof_to_obs_smpl)."""
    fn = refload.env_functions()
    g = syn.make_generator(4242)
    rb = syn.rigid_body_state(g, n)
    dof_pos = torch.randn(n, 69, generator=g) * 0.7
    dof_pos[0, 0:3] = 0.0                                   # zero exp-map -> masked branch
    dof_pos[1, 3:6] = torch.tensor([0.0, 0.0, 1e-6])
    dof_vel = torch.randn(n, 69, generator=g)
    key = [7, 3, 22, 17]                                    # R_Ankle, L_Ankle, R_Wrist, L_Wrist (env_im.yaml:35)
    bp, br, bv, ba = rb[..., 0:3], rb[..., 3:7], rb[..., 7:10], rb[..., 10:13]
    empty = torch.zeros(n, 0)
    none_subset = torch.zeros(0, dtype=torch.long)
    full = fn["build_amp_observations_smpl"](bp[:, 0], br[:, 0], bv[:, 0], ba[:, 0], dof_pos, dof_vel, bp[:, key], empty, empty,
                                             none_subset, True, True, False, False, False, True)
    joints19 = [j for j in range(23) if j not in (3, 7, 17, 22)]        # drop toes / hands (dofs 9:12, 21:24, 51:54, 66:69)
    subset = torch.tensor([3 * j + k for j in joints19 for k in range(3)], dtype=torch.long)
    sub = fn["build_amp_observations_smpl"](bp[:, 0], br[:, 0], bv[:, 0], ba[:, 0], dof_pos, dof_vel, bp[:, key], empty, empty,
                                            subset, True, False, True, False, False, True)
    glob = fn["build_amp_observations_smpl"](bp[:, 0], br[:, 0], bv[:, 0], ba[:, 0], dof_pos, dof_vel, bp[:, key], empty, empty,
                                             none_subset, False, True, False, False, False, True)
    out = {"rb": rb, "dof_pos": dof_pos, "dof_vel": dof_vel, "key_body_ids": np.array(key), "joints19": np.array(joints19),
           "amp_obs_full": full, "amp_obs_subset19_noheight": sub, "amp_obs_global_root": glob,
           "dof_to_obs": fn["dof_to_obs_smpl"](dof_pos)}
    np.savez_compressed(os.path.join(GOLDEN_DIR, "env_amp.npz"), **_np(out))


def gen_agent_math():
    m = refload.agent_methods()
    g = syn.make_generator(4321)
    out = {}
    for tag, (t, n) in {"a": (16, 64), "b": (32, 67)}.items():
        rewards, values, next_values, dones = syn.rollout_scalars(g, t, n, done_p=0.05)
        if tag == "b":
            dones[:, 3] = 1      # all-done column
            dones[-1, :] = 1     # every env ends on the last step
        me = refload.stub_self(horizon_length=t, gamma=0.99, tau=0.95)
        advs = m["discount_values"](me, dones.float(), values, rewards, next_values)
        out.update({f"gae_{tag}_rewards": rewards, f"gae_{tag}_values": values,
                    f"gae_{tag}_next_values": next_values, f"gae_{tag}_dones": dones,
                    f"gae_{tag}_advs": advs})
    b, a = 515, 69
    old_nlp = torch.randn(b, generator=g) * 0.5 + 60
    nlp = old_nlp + 0.3 * torch.randn(b, generator=g)
    adv = torch.randn(b, generator=g)
    me = refload.stub_self(bounds_loss_coef=10.0, normalize_advantage=True)
    ai = m["_actor_loss"](me, old_nlp, nlp, adv, 0.2)
    vp = torch.randn(b, 1, generator=g)
    val = vp + 0.4 * torch.randn(b, 1, generator=g)
    ret = torch.randn(b, 1, generator=g)
    c_noclip = m["_critic_loss"](me, vp, val, 0.2, ret, False)["critic_loss"]
    c_clip = m["_critic_loss"](me, vp, val, 0.2, ret, True)["critic_loss"]
    mu = 1.2 * torch.randn(b, a, generator=g)
    bl = m["bound_loss"](me, mu)
    returns = torch.randn(b, 1, generator=g)
    values = torch.randn(b, 1, generator=g)
    advs_n = m["_calc_advs"](me, {"returns": returns, "values": values})
    out.update({"loss_old_neglogp": old_nlp, "loss_neglogp": nlp, "loss_adv": adv,
                "actor_loss": ai["actor_loss"], "actor_clipped": ai["actor_clipped"],
                "loss_old_values": vp, "loss_values": val, "loss_returns": ret,
                "critic_loss": c_noclip, "critic_loss_clipped": c_clip,
                "loss_mu": mu, "bound_loss": bl,
                "advs_returns": returns, "advs_values": values, "advs_normalized": advs_n})
    np.savez_compressed(os.path.join(GOLDEN_DIR, "agent_math.npz"), **_np(out))


def gen_rms():
    mods = refload.importable_modules()
    RMS = mods["running_mean_std"].RunningMeanStd
    g = syn.make_generator(99)
    f = 37
    rms = RMS((f,))
    rms.train()
    out = {}
    scale = 1.0 + 4.0 * torch.rand(f, generator=g)
    shift = 3.0 * torch.randn(f, generator=g)
    for i, b in enumerate([64, 33, 128]):
        x = torch.randn(b, f, generator=g) * scale + shift
        y = rms(x)
        out[f"x{i}"] = x
        out[f"y{i}"] = y
        out[f"mean{i}"] = rms.running_mean.clone()
        out[f"var{i}"] = rms.running_var.clone()
        out[f"count{i}"] = rms.count.clone()
    rms.eval()
    x = torch.randn(16, f, generator=g) * scale + shift
    out["x_eval"] = x
    out["y_eval"] = rms(x)
    out["y_unnorm"] = rms(torch.randn(16, f, generator=g) * 3.0, unnorm=True)
    out["x_unnorm_in"] = torch.zeros(0)
    # re-create the unnorm input deterministically for the consumer
    g2 = syn.make_generator(100)
    z = torch.randn(16, f, generator=g2) * 3.0
    out["z_unnorm_in"] = z
    out["z_unnorm_out"] = rms(z, unnorm=True)
    # frozen copy must not update
    rms.train()
    rms.freeze()
    before = rms.running_mean.clone()
    _ = rms(x)
    assert torch.equal(before, rms.running_mean)
    kl = mods["loss_functions"].kl_multi
    qm, qv, pm, pv = (torch.randn(50, 32, generator=g) for _ in range(4))
    out.update({"kl_qm": qm, "kl_qv": qv, "kl_pm": pm, "kl_pv": pv, "kl_multi": kl(qm, qv, pm, pv)})
    np.savez_compressed(os.path.join(GOLDEN_DIR, "rms.npz"), **_np(out))


def gen_motion_lib(num_motions=9, n=203):
    """MotionLibBase.get_motion_state / get_root_pos_smpl / _calc_frame_blend / get_motion_num_steps run from the reference's own
    source on a synthetic library (tables stored alongside so the test needs nothing else)."""
    cls = refload.motion_lib_class()
    g = syn.make_generator(777)
    tabs = syn.synthetic_motion_library(g, num_motions, min_frames=8, max_frames=40)
    ref = cls()
    for k in ("gts", "grs", "lrs", "gvs", "gavs", "dvs", "length_starts"):
        setattr(ref, k, tabs[k])
    ref._motion_lengths, ref._motion_fps, ref._motion_dt = tabs["motion_lengths"], tabs["motion_fps"], tabs["motion_dt"]
    ref._motion_num_frames, ref.num_bodies, ref._device = tabs["motion_num_frames"], syn.NUM_BODIES, "cpu"
    ref._motion_aa = torch.zeros(tabs["gts"].shape[0], 72)
    ref._motion_bodies = torch.zeros(num_motions, 17)
    ref._motion_limb_weights = torch.zeros(num_motions, 10)
    ids = torch.randint(0, num_motions, (n,), generator=g)
    length = tabs["motion_lengths"][ids]
    times = torch.rand(n, generator=g) * length * 1.15 - 0.05 * length          # some before 0 and past the end
    dt = tabs["motion_dt"][ids]
    k = torch.randint(0, 8, (n,), generator=g).float()
    exact = torch.rand(n, generator=g) < 0.3                                     # exact frame times (blend 0 / float edge)
    times = torch.where(exact, torch.minimum(k * dt, length), times)
    times[0], times[1], times[2] = 0.0, length[1], -0.3
    offset = torch.randn(n, 3, generator=g) * torch.tensor([2.0, 2.0, 0.0])
    res = ref.get_motion_state(ids, times, offset)
    res_no = ref.get_motion_state(ids, times)
    f0, f1, blend = ref._calc_frame_blend(times, length, tabs["motion_num_frames"][ids], dt)
    out = {"tab_" + k: v for k, v in tabs.items()}
    out.update({"motion_ids": ids, "motion_times": times, "offset": offset, "frame_idx0": f0, "frame_idx1": f1, "blend": blend,
                "root_pos_smpl": ref.get_root_pos_smpl(ids, times)["root_pos"], "num_steps": ref.get_motion_num_steps(),
                "rg_pos_no_offset": res_no["rg_pos"]})
    for key in ("root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "rg_pos", "rb_rot", "body_vel", "body_ang_vel"):
        out[key] = res[key]
    np.savez_compressed(os.path.join(GOLDEN_DIR, "motion_lib.npz"), **_np(out))


def gen_tasks(n=61):
    """Downstream-task functions (speed / reach / strike + the base reset) from the reference's own TorchScript sources."""
    fn = refload.task_functions()
    g = syn.make_generator(555)
    rb = syn.rigid_body_state(g, n)
    root = rb[:, 0].clone()
    prev_root = root[:, 0:3] + 0.02 * torch.randn(n, 3, generator=g)
    tar_speed = 1.0 + 2.0 * torch.rand(n, generator=g)
    tar_pos = root[:, 0:3] + torch.randn(n, 3, generator=g)
    tar_states = syn.rigid_body_state(g, n)[:, 0].clone()
    tar_states[:5, 3:7] = torch.tensor([0.0, 0.0, 0.0, 1.0])                 # upright targets (tar_rot_err = 1) ...
    tar_states[5:10, 3:7] = torch.tensor([1.0, 0.0, 0.0, 0.0])               # ... and knocked-over ones (success branch)
    contact = torch.randn(n, 24, 3, generator=g) * (torch.rand(n, 24, 1, generator=g) < 0.3)
    contact[::4] *= 100.0
    tar_contact = torch.randn(n, 3, generator=g) * 60.0
    progress = torch.randint(0, 305, (n,), generator=g)
    progress[:3] = torch.tensor([0, 1, 2])
    progress[-3:] = torch.tensor([298, 299, 304])                            # time-outs (>= max_episode_length - 1) ...
    contact[-3:] = 0.0                                                       # ... on envs that have not fallen
    body_pos = rb[..., 0:3].clone()
    body_pos[::3, 5, 2] = 0.05                                               # a body below its termination height
    term_h = torch.full((24,), 0.15)
    contact_ids = torch.tensor([7, 3, 8, 4])
    strike_ids = torch.tensor([23, 22, 21])
    reset0 = torch.zeros(n, dtype=torch.long)
    dt = 1.0 / 30.0
    out = {"root_states": root, "prev_root_pos": prev_root, "tar_speed": tar_speed, "tar_pos": tar_pos, "tar_states": tar_states, "contact": contact,
           "tar_contact": tar_contact, "progress": progress, "body_pos": body_pos, "term_h": term_h, "contact_ids": contact_ids, "strike_ids": strike_ids,
           "reach_body_pos": rb[:, 23, 0:3].clone(), "strike_body_vel": rb[:, 23, 7:10].clone(), "dt": torch.tensor(dt)}
    out["speed_obs"] = fn["compute_speed_observations"](root, tar_speed)
    out["speed_rew"] = fn["compute_speed_reward"](root[:, 0:3], prev_root, root[:, 3:7], tar_speed, dt)
    out["loc_obs"] = fn["compute_location_observations"](root, tar_pos)
    out["reach_rew"] = fn["compute_reach_reward"](out["reach_body_pos"], root[:, 3:7], tar_pos, 1.0, dt)
    out["strike_obs"] = fn["compute_strike_observations"](root, tar_states)
    out["strike_rew"] = fn["compute_strike_reward"](tar_states[:, 0:3], tar_states[:, 3:7], root, prev_root, out["strike_body_vel"], dt, 1.4)
    out["reset"], out["terminated"] = fn["compute_humanoid_reset"](reset0, progress, contact, contact_ids, body_pos, 300.0, True, term_h)
    out["reset_noearly"], _ = fn["compute_humanoid_reset"](reset0, progress, contact, contact_ids, body_pos, 300.0, False, term_h)
    out["strike_reset"], out["strike_terminated"] = fn["strike_compute_humanoid_reset"](reset0, progress, contact, contact_ids, body_pos, tar_contact,
                                                                                    strike_ids, 300.0, True, term_h)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "tasks.npz"), **_np(out))


def gen_terrain(n=37):
    """HumanoidTraj / HumanoidPedestrianTerrain (terrain traversal): the reference's TrajGenerator (same seed -> same trajectories), its
    TorchScript functions and the get_heights / get_center_heights / sample_height_points methods executed on a stub object."""
    T = refload.terrain_functions()
    g = syn.make_generator(909)
    num_verts, episode_dur, dt = 101, 300 * (2.0 / 60.0), 2.0 / 60.0
    rb = syn.rigid_body_state(g, n)
    rb[:, :, 0:2] += 12.0 + 6.0 * torch.rand(n, 1, 2, generator=g)                   # inside the map (cells of 0.1 m, 260 x 300)
    root = rb[:, 0].clone()
    head = rb[:, 13].clone()
    tg = T["TrajGenerator"](n, episode_dur, num_verts, "cpu", 2.0, 0.0, 3.0, 2.0, 0.02)
    torch.manual_seed(4321)
    tg.reset(torch.arange(n), root[:, 0:3])
    verts = tg._verts.clone()
    torch.manual_seed(4321)                                                            # the same draws, in the reference's order
    u = {"dtheta": torch.rand(n, num_verts - 1), "sharp": torch.rand(n, num_verts - 1)}
    u["sharp_mask"] = torch.bernoulli(0.02 * torch.ones(n, num_verts - 1)) == 1.0
    u["heading"], u["dspeed"], u["speed0"] = torch.rand(n), torch.rand(n, num_verts - 1), torch.rand(n)
    progress = torch.randint(0, 300, (n,), generator=g)
    progress[:3] = torch.tensor([0, 1, 299])
    stub = refload.stub_self(num_envs=n, device="cpu", dt=dt, progress_buf=progress, _num_traj_samples=10, _traj_sample_timestep=0.5, _traj_gen=tg)
    samples = T["HumanoidTraj._fetch_traj_samples"](stub)
    times = progress * dt
    tar_pos = tg.calc_pos(torch.arange(n), times)
    hs = syn.synthetic_height_field()
    sensor_res, ext = 32, 2.0
    yy = torch.tensor(np.linspace(-ext, ext, sensor_res))
    xx = torch.tensor(np.linspace(-ext, ext, sensor_res))
    gx, gy = torch.meshgrid(xx, yy)
    hp = torch.zeros(n, sensor_res * sensor_res, 3)
    hp[:, :, 0], hp[:, :, 1] = gx.flatten(), gy.flatten()
    cy, cx = torch.tensor(np.linspace(-0.2, 0.2, 3)), torch.tensor(np.linspace(-0.1, 0.1, 3))
    cgx, cgy = torch.meshgrid(cx, cy)
    cp = torch.zeros(n, 9, 3)
    cp[:, :, 0], cp[:, :, 1] = cgx.flatten(), cgy.flatten()
    terrain = refload.stub_self(heightsamples=hs, horizontal_scale=0.1, vertical_scale=0.005)
    terrain.world_points_to_map = lambda pts: T["Terrain.world_points_to_map"](terrain, pts)
    terrain.sample_height_points = lambda pts, **kw: T["Terrain.sample_height_points"](terrain, pts, **kw)
    out = {"rb": rb, "verts": verts, "progress": progress, "traj_samples": samples, "tar_pos": tar_pos, "heightsamples": hs,
           "height_points": hp[0, :, 0:2].clone(), "center_points": cp[0, :, 0:2].clone(), "dt": torch.tensor(dt),
           "traj_dt": torch.tensor(episode_dur / (num_verts - 1)), **{"u_" + k: v for k, v in u.items()}}
    for up in (True, False):
        tag = "" if up else "_noup"
        env = refload.stub_self(cfg={"env": {"terrain": {"terrainType": "trimesh"}}}, num_envs=n, device="cpu", humanoid_type="smpl", _has_upright_start=up,
                                num_height_points=sensor_res * sensor_res, num_center_height_points=9, height_points=hp, center_height_points=cp,
                                velocity_map=False, _divide_group=False, _group_obs=False, _disable_group_obs=False, terrain=terrain,
                                _humanoid_root_states=root)
        heights = T["HumanoidPedestrianTerrain.get_heights"](env, root_states=head[:, 0:7], env_ids=None)
        center = T["HumanoidPedestrianTerrain.get_center_heights"](env, root_states=root, env_ids=None)
        out[f"heights{tag}"], out[f"center_heights{tag}"] = heights, center
        out[f"loc_obs{tag}"] = T["compute_location_observations"](root, samples, up)
        hobs = torch.clip(center.mean(dim=-1, keepdim=True) - heights, -3, 3.) * 5                     # _compute_task_obs :414-424
        out[f"task_obs{tag}"] = torch.cat([out[f"loc_obs{tag}"], hobs], dim=1)
    out["traj_loc_obs"] = T["traj_compute_location_observations"](root, samples)
    out["loc_rew"] = T["compute_location_reward"](root[:, 0:3], tar_pos)
    out["loc_rew_fuzzy"] = T["compute_location_reward_fuzzy"](root[:, 0:3], tar_pos + 0.03)
    contact = torch.zeros(n, 24, 3)
    contact[::3, 7] = torch.tensor([10.0, 0.0, 400.0])                       # a foot in contact (ignored)
    contact[1::4, 12] = torch.tensor([30.0, 30.0, 30.0])                     # a non-foot contact above 50 N in norm
    contact[2::5, 16] = torch.tensor([0.3, 0.0, 0.0])                        # a light non-foot contact (fall_contact of the traj variant, not of the terrain one)
    contact_ids = torch.tensor([7, 3, 8, 4])
    far = tar_pos.clone()
    far[5::6, 0] += 5.0                                                      # too far from the trajectory
    term_h = torch.full((24,), 0.15)
    bp = rb[..., 0:3].clone()
    bp[2::5, 16, 2] = 0.05
    reset0 = torch.zeros(n, dtype=torch.long)
    out.update({"contact": contact, "contact_ids": contact_ids, "far_tar_pos": far, "term_h": term_h, "body_pos": bp})
    out["terrain_reset"], out["terrain_terminated"] = T["terrain_compute_humanoid_reset"](reset0, progress, contact, contact_ids, torch.zeros(n), bp, far,
                                                                                        300.0, 4.0, True, term_h, False)
    out["terrain_reset_noearly"], _ = T["terrain_compute_humanoid_reset"](reset0, progress, contact, contact_ids, torch.zeros(n), bp, far, 300.0, 4.0, False,
                                                                          term_h, False)
    out["traj_reset"], out["traj_terminated"] = T["traj_compute_humanoid_reset"](reset0, progress, contact, contact_ids, bp, far, 300.0, 4.0, True, term_h)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "terrain.npz"), **_np(out))


def zero_out_far_inputs(n=83, seed=2025):
    """One post-physics step of a ``zero_out_far: True`` env (phc_kp_pnn_iccv.yaml:36): whole humanoids standing 0 .. 8 m off their
    reference (inside close_distance, between close and far, beyond far_distance), _point_goal of the step before."""
    g = syn.make_generator(seed)
    d = syn.env_step_inputs(g, n)
    shift = torch.zeros(n, 2)
    k = n // 4
    shift[k:2 * k, 0] = torch.linspace(0.26, 2.9, k)                 # between close_distance and far_distance: own state as reference
    ang = torch.rand(n - 2 * k, generator=g) * 6.2831853
    rad = torch.linspace(3.05, 8.0, n - 2 * k)                       # beyond far_distance: the root target is a direction
    shift[2 * k:, 0], shift[2 * k:, 1] = rad * torch.cos(ang), rad * torch.sin(ang)
    shift[2] = torch.tensor([0.2, 0.1])                              # |.| ~ 0.224 + noise: around the 0.25 m transition of the reward
    shift[3] = torch.tensor([0.15, 0.2])
    d["rb"][:, :, 0:2] += shift[:, None, :]
    dist_now = torch.norm(d["rb"][:, 0, 0:3] - d["ref_now"]["pos"][:, 0], dim=-1)
    prev = dist_now + 0.05 * torch.randn(n, generator=g)
    prev[::5] += 0.5                                                 # approached by more than 1/3 m: the clamp of compute_point_goal_reward
    d["point_goal"] = prev
    return d


def gen_env_zero_out_far():
    """HumanoidIm._compute_reward / _compute_reset / _compute_task_obs METHOD BODIES with zero_out_far (humanoid_im.py:763-777, 814-826,
    870-887, 1158-1176) and compute_point_goal_reward (:1577-1582), run on a stub task whose motion cache hands out recorded frames."""
    import types
    f = refload.humanoid_im_methods()
    fn = refload.env_functions()
    d = zero_out_far_inputs()
    rb, n = d["rb"], d["rb"].shape[0]
    rn, rx = d["ref_now"], d["ref_next"]
    out = {"rb": rb, "dof_force": d["dof_force"], "dof_vel": d["dof_vel"], "progress": d["progress"], "pass_time": d["pass_time"],
           "point_goal_prev": d["point_goal"], "close_distance": torch.tensor(0.25), "far_distance": torch.tensor(3.0)}
    for k in ("pos", "rot", "vel", "ang"):
        out[f"ref_now_{k}"], out[f"ref_next_{k}"] = rn[k], rx[k]
    out["point_goal_reward"], _ = fn["compute_point_goal_reward"](d["point_goal"], torch.norm(rb[:, 0, 0:3] - rn["pos"][:, 0], dim=-1))
    dt = 1.0 / 30

    def res(r):
        return {"root_pos": r["pos"][:, 0].clone(), "root_rot": r["rot"][:, 0].clone(), "dof_pos": torch.zeros(n, 69), "root_vel": r["vel"][:, 0].clone(),
                "root_ang_vel": r["ang"][:, 0].clone(), "dof_vel": torch.zeros(n, 69), "motion_bodies": torch.zeros(n, 17),
                "motion_limb_weights": torch.zeros(n, 10), "motion_aa": torch.zeros(n, 72), "rg_pos": r["pos"].clone(), "rb_rot": r["rot"].clone(),
                "body_vel": r["vel"].clone(), "body_ang_vel": r["ang"].clone()}

    for obs_v, ids, tag in ((6, list(range(24)), "v6"), (7, syn.VR_TRACK_BODY_IDS, "v7_vr"), (7, list(range(24)), "v7"), (8, list(range(24)), "v8"),
                            (9, list(range(24)), "v9"), (6, syn.VR_TRACK_BODY_IDS, "v6_vr")):
        for close, far, dtag in ((0.25, 3.0, ""), (0.5, 1.5, "_c05_f15")):
            task = types.SimpleNamespace(
                _rigid_body_pos=rb[..., 0:3], _rigid_body_rot=rb[..., 3:7], _rigid_body_vel=rb[..., 7:10], _rigid_body_ang_vel=rb[..., 10:13],
                num_envs=n, device="cpu", humanoid_shapes=torch.zeros(n, 17), _fut_tracks=False, _num_traj_samples=1, _traj_sample_timestep=1.0 / 30,
                progress_buf=d["progress"].clone(), dt=dt, _motion_start_times=torch.zeros(n), _motion_start_times_offset=torch.zeros(n),
                _sampled_motion_ids=torch.arange(n), _global_offset=torch.zeros(n, 3), _track_bodies_id=torch.tensor(ids), obs_v=obs_v,
                _has_upright_start=True, zero_out_far=True, zero_out_far_train=True, close_distance=close, far_distance=far,
                _point_goal=d["point_goal"].clone(), _occl_training=False, _fut_tracks_dropout=False, ref_body_pos=torch.zeros(n, 24, 3),
                ref_body_vel=torch.zeros(n, 24, 3), ref_body_rot=torch.zeros(n, 24, 4), ref_body_pos_subset=torch.zeros(n, len(ids), 3),
                ref_dof_pos=torch.zeros(n, 69), dof_force_tensor=d["dof_force"], _dof_vel=d["dof_vel"],
                reward_specs={"k_pos": 100, "k_rot": 10, "k_vel": 0.1, "k_ang_vel": 0.1, "w_pos": 0.5, "w_rot": 0.3, "w_vel": 0.1, "w_ang_vel": 0.1},
                _full_body_reward=True, power_reward=True, power_coefficient=0.0005, rew_buf=torch.zeros(n), reward_raw=torch.zeros(n, 4),
                max_episode_length=300, cycle_motion=False, cycle_motion_xp=False, _cycle_counter=torch.zeros(n, dtype=torch.int64),
                reset_buf=torch.zeros(n, dtype=torch.int64), _terminate_buf=torch.zeros(n, dtype=torch.int64), _contact_forces=torch.zeros(n, 24, 3),
                _contact_body_ids=torch.tensor([7, 3]), _reset_bodies_id=torch.tensor(syn.RESET_BODY_IDS), _enable_early_termination=True,
                _termination_distances=torch.full((24,), 0.25), strict_eval=False,
                _motion_lib=types.SimpleNamespace(_motion_lengths=torch.where(d["pass_time"], torch.zeros(n), torch.full((n,), 1e9))))
            for k, m in f.items():
                setattr(task, k, types.MethodType(m, task))
            which = {"r": rn}
            task._get_state_from_motionlib_cache = lambda ids_, times_, offset_=None: res(which["r"])
            task._compute_reward(None)
            task._compute_reset()
            which["r"] = rx
            obs = task._compute_task_obs()
            out[f"task_obs_{tag}{dtag}"], out[f"point_goal_{tag}{dtag}"] = obs, task._point_goal.clone()
            if tag == "v6" and not dtag:
                out["reward"], out["reward_raw"] = task.rew_buf.clone(), task.reward_raw.clone()
                out["reset"], out["terminate"] = task.reset_buf.clone(), task._terminate_buf.clone()
    np.savez_compressed(os.path.join(GOLDEN_DIR, "env_zero_out_far.npz"), **_np(out))


def main():
    assert refload.available(), "reference tree not found; goldens can only be generated in the build container"
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(1)
    gen_rotations()
    gen_env_im()
    gen_env_amp()
    gen_env_variants()
    gen_env_shape_obs()
    gen_env_zero_out_far()
    gen_terrain()
    gen_agent_math()
    gen_rms()
    gen_motion_lib()
    gen_tasks()
    for f in sorted(os.listdir(GOLDEN_DIR)):
        print(f, os.path.getsize(os.path.join(GOLDEN_DIR, f)), "bytes")


if __name__ == "__main__":
    main()
