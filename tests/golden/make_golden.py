#!/usr/bin/env python3
"""Generate the golden vectors in this directory from the REAL reference.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden.py

It imports /root/reference/model/{unet,view_fusion}.py on the CPU (fp32), drives them
with seeded inputs and writes small .npz files holding INPUTS (or the seeds that make
them) and EXPECTED OUTPUTS.  Weights are never stored: both sides re-create them with
`view_fusion_amd.utils.deterministic_fill_` (keyed on state_dict order).

Vectors:
  G0 state_dict_keys.json   ordered (key, shape) list of ViewFusion.state_dict(), tiny + small
  G1 schedules.npz      six schedule buffers for 8 (schedule, T, start, end) settings
  G2 unet_tiny.npz      tiny UNet fwd output + digests of every parameter gradient
     unet_small.npz     small-64x64 config (the real 33.9 M-param net) fwd output, S=2
  G3 train_*.npz        ViewFusion.forward loss + gradient digests; pinned t,u,noise;
                        uniform / ragged view_count, weighting on / off
  G4 sample_*.npz       p_mean_variance / generate() chains with injected noise
  G5 c1_small_chain.npz BASELINE config C1: small UNet, B=2 N=2, 10-step chain + one training fwd/bwd (digests)
  G6 train_relative.npz `relative` conditioning (in_channel 9, 6-channel views): loss, grads, p_mean_variance
  G7 unet_tiny_dropout.npz  UNet(dropout=0.1) training-mode fwd/bwd with the recorded Dropout masks
  G8 ckpt_ref_micro.pt + ckpt_continue.npz   checkpoint written by the reference's Checkpoint.save after one Adam
                        step, and the parameters after two more steps
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from model.unet import UNet                      # noqa: E402  (reference)
from model.view_fusion import ViewFusion         # noqa: E402  (reference)
from view_fusion_amd.utils import deterministic_fill_, tensor_digest  # noqa: E402

torch.set_num_threads(8)

TINY = dict(in_channel=6, out_channel=6, inner_channel=32, norm_groups=32,
            channel_mults=(1, 2), attn_res=(8,), res_blocks=1, image_size=16)
SMALL = dict(in_channel=6, out_channel=6, inner_channel=64, norm_groups=32,
             channel_mults=(1, 2, 3, 5), attn_res=(16,), res_blocks=3, image_size=64)

SCHEDULES = {
    "linear_train": dict(schedule="linear", num_timesteps=2000, linear_start=1e-6, linear_end=1e-2),
    "linear_test": dict(schedule="linear", num_timesteps=1000, linear_start=1e-4, linear_end=0.09),
    "quad": dict(schedule="quad", num_timesteps=10, linear_start=1e-4, linear_end=0.09),
    "warmup10": dict(schedule="warmup10", num_timesteps=20, linear_start=1e-4, linear_end=0.09),
    "warmup50": dict(schedule="warmup50", num_timesteps=10, linear_start=1e-4, linear_end=0.09),
    "const": dict(schedule="const", num_timesteps=10, linear_start=1e-4, linear_end=0.09),
    "jsd": dict(schedule="jsd", num_timesteps=10),
    "cosine": dict(schedule="cosine", num_timesteps=10),
}
BUFS = ("gammas", "sqrt_recip_gammas", "sqrt_recipm1_gammas", "posterior_log_variance_clipped",
        "posterior_mean_coef1", "posterior_mean_coef2")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def make_vf(hp, sched, weighting=True):
    net = UNet(**hp)
    deterministic_fill_(net.state_dict())        # keyed on the UNet's own state_dict order
    vf = quiet(ViewFusion, net, {"train": sched}, weighting, weighting)
    vf.set_new_noise_schedule(device=torch.device("cpu"), phase="train")
    return vf


def grads_digest(module):
    out = {}
    for k, p in module.named_parameters():
        d = tensor_digest(p.grad)
        out[f"g.{k}.stat"] = np.array([d["sum"], d["l2"], d["absmax"]])
        out[f"g.{k}.samples"] = d["samples"]
    return out


def inputs(B, N, hw, seed):
    g = torch.Generator().manual_seed(seed)
    y_0 = torch.rand(B, 3, hw, hw, generator=g)
    y_cond = torch.rand(B, N, 3, hw, hw, generator=g)
    angle = 2 * np.pi / 24 * torch.randint(0, 24, (B, 1), generator=g).float()
    noise = torch.randn(B, 3, hw, hw, generator=g)
    return y_0, y_cond, angle, noise


def g0():
    """state_dict key order + shapes of the reference modules (the serialized contract)."""
    import json
    out = {}
    for tag, hp in (("tiny", TINY), ("small", SMALL)):
        vf = make_vf(hp, SCHEDULES["linear_train"], True)
        out[tag] = [[k, list(v.shape)] for k, v in vf.state_dict().items()]
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(out, f)


def g1():
    out = {}
    for name, kw in SCHEDULES.items():
        vf = quiet(ViewFusion, None, {"train": kw})
        vf.set_new_noise_schedule(device=torch.device("cpu"), phase="train")
        for b in BUFS:
            out[f"{name}.{b}"] = getattr(vf, b).numpy()
    np.savez_compressed(os.path.join(HERE, "schedules.npz"), **out)


def g2():
    for tag, hp, S, hw in (("tiny", TINY, 3, 16), ("small", SMALL, 2, 64)):
        net = UNet(**hp)
        deterministic_fill_(net.state_dict())
        g = torch.Generator().manual_seed(7)
        x = torch.rand(S, 6, hw, hw, generator=g)
        angle = 2 * np.pi / 24 * torch.randint(0, 24, (S, 1), generator=g).float()
        level = torch.rand(S, 1, generator=g)
        out = dict(x=x.numpy(), angle=angle.numpy(), level=level.numpy())
        if tag == "tiny":
            x.requires_grad_(True)
            y = net(x, angle, level)
            gy = torch.randn(y.shape, generator=g)
            (y * gy).sum().backward()
            out.update(y=y.detach().numpy(), gy=gy.numpy(), gx=x.grad.numpy())
            out.update(grads_digest(net))
        else:
            with torch.no_grad():
                y = net(x, angle, level)
            out.update(y=y.numpy())
        np.savez_compressed(os.path.join(HERE, f"unet_{tag}.npz"), **out)


def g3():
    sched = SCHEDULES["linear_train"]
    cases = {
        "uniform_w": dict(B=2, N=2, vc=[2, 2], weighting=True),
        "ragged_w": dict(B=3, N=3, vc=[1, 3, 2], weighting=True),
        "ragged_mean": dict(B=3, N=3, vc=[1, 3, 2], weighting=False),
    }
    for tag, c in cases.items():
        vf = make_vf(TINY, sched, c["weighting"])
        y_0, y_cond, angle, noise = inputs(c["B"], c["N"], 16, seed=11)
        vc = torch.tensor(c["vc"], dtype=torch.long)
        # the reference draws t = randint(1,T,(b,)) then rand((b,1)) from the global CPU
        # generator (view_fusion.py:231-237): replay the same two draws to record them.
        torch.manual_seed(123)
        t = torch.randint(1, vf.num_timesteps, (c["B"],)).long()
        u = torch.rand((c["B"], 1))
        torch.manual_seed(123)
        loss = vf(y_cond=y_cond, view_count=vc, angle=angle, y_0=y_0, noise=noise)
        loss.backward()
        out = dict(y_0=y_0.numpy(), y_cond=y_cond.numpy(), angle=angle.numpy(), noise=noise.numpy(),
                   view_count=vc.numpy(), t=t.numpy(), u=u.numpy(), loss=np.float64(loss.item()),
                   weighting=np.array(c["weighting"]))
        out.update(grads_digest(vf.denoise_fn))
        np.savez_compressed(os.path.join(HERE, f"train_{tag}.npz"), **out)


def g4():
    # (a) generate(): T=10 so that T > sample_num=8 and every step is stashed
    for tag, weighting in (("w", True), ("mean", False)):
        sched = dict(schedule="linear", num_timesteps=10, linear_start=1e-4, linear_end=0.09)
        vf = make_vf(TINY, sched, weighting)
        _, y_cond, angle, _ = inputs(2, 2, 16, seed=21)
        vc = torch.tensor([2, 1], dtype=torch.long)
        g = torch.Generator().manual_seed(5)
        y_T = torch.randn(2, 3, 16, 16, generator=g)
        # p_sample draws randn_like(y_t) for every step with t>0 (view_fusion.py:176)
        torch.manual_seed(77)
        z = [torch.randn(2, 3, 16, 16) for _ in range(9)]           # steps i = 9 .. 1
        z_seq = torch.stack([torch.zeros(2, 3, 16, 16)] + z[::-1])   # z_seq[i] = noise of step i
        torch.manual_seed(77)
        y, ret, logit_arr, weight_arr, samples = quiet(vf.generate, y_cond, vc, angle, y_t=y_T)
        out = dict(y_cond=y_cond.numpy(), angle=angle.numpy(), view_count=vc.numpy(), y_T=y_T.numpy(),
                   z_seq=z_seq.numpy(), y=y.numpy(), ret=ret.numpy(), samples=samples.numpy())
        if weighting:
            out.update(logit_arr=logit_arr.numpy(), weight_arr=weight_arr.numpy())
        np.savez_compressed(os.path.join(HERE, f"sample_generate_{tag}.npz"), **out)

    # (b) one p_mean_variance call at a mid-chain t with the real T=1000 sampler schedule
    vf = make_vf(TINY, SCHEDULES["linear_test"], True)
    _, y_cond, angle, _ = inputs(3, 3, 16, seed=31)
    vc = torch.tensor([3, 1, 2], dtype=torch.long)
    g = torch.Generator().manual_seed(6)
    y_t = torch.randn(3, 3, 16, 16, generator=g)
    t = torch.tensor([999, 500, 0], dtype=torch.long)
    with torch.no_grad():
        mean, logvar, logits, w = vf.p_mean_variance(y_t, y_cond, vc, angle, t, clip_denoised=True)
    np.savez_compressed(os.path.join(HERE, "sample_pmv.npz"), y_cond=y_cond.numpy(), angle=angle.numpy(),
                        view_count=vc.numpy(), y_t=y_t.numpy(), t=t.numpy(), mean=mean.numpy(),
                        logvar=logvar.numpy(), logits=logits.numpy(), weights=w.numpy())


MICRO = dict(in_channel=6, out_channel=6, inner_channel=8, norm_groups=8,
             channel_mults=(1, 2), attn_res=(8,), res_blocks=1, image_size=16)      # 41 k params: checkpoint fixture
C1_SCHED = dict(schedule="linear", num_timesteps=10, linear_start=1e-4, linear_end=0.09)


def draw_t_u(vf, B, seed):
    """Replay the two draws ViewFusion.forward makes from the global CPU generator (view_fusion.py:231-237)."""
    torch.manual_seed(seed)
    t = torch.randint(1, vf.num_timesteps, (B,)).long()
    u = torch.rand((B, 1))
    torch.manual_seed(seed)
    return t, u


def g5():
    """BASELINE config C1 as stated: configs/small-v100.yaml UNet (33.9 M params) on CPU, B=2 N=2 64x64, a 10-step
    DDPM chain with injected y_T / z, plus one training forward+backward on the same shapes."""
    vf = make_vf(SMALL, C1_SCHED, True)
    y_0, y_cond, angle, noise = inputs(2, 2, 64, seed=41)
    vc = torch.tensor([2, 2], dtype=torch.long)
    g = torch.Generator().manual_seed(42)
    y_T = torch.randn(2, 3, 64, 64, generator=g)
    torch.manual_seed(43)
    z = [torch.randn(2, 3, 64, 64) for _ in range(9)]
    z_seq = torch.stack([torch.zeros(2, 3, 64, 64)] + z[::-1])
    torch.manual_seed(43)
    y, ret, logit_arr, weight_arr, samples = quiet(vf.generate, y_cond, vc, angle, y_t=y_T)
    out = dict(seed_inputs=np.array(41), seed_yT=np.array(42), seed_z=np.array(43), view_count=vc.numpy(),
               y=y.numpy(), samples=samples.numpy())
    for name, tns in (("ret", ret), ("logit_arr", logit_arr), ("weight_arr", weight_arr)):
        d = tensor_digest(tns, nsamples=256)
        out[f"{name}.stat"] = np.array([d["sum"], d["l2"], d["absmax"]])
        out[f"{name}.samples"] = d["samples"]
        out[f"{name}.shape"] = np.array(tns.shape)
    # training iteration on the same shapes (train schedule T=2000)
    vf = make_vf(SMALL, SCHEDULES["linear_train"], True)
    t, u = draw_t_u(vf, 2, seed=44)
    loss = vf(y_cond=y_cond, view_count=vc, angle=angle, y_0=y_0, noise=noise)
    loss.backward()
    out.update(t=t.numpy(), u=u.numpy(), loss=np.float64(loss.item()))
    out.update(grads_digest(vf.denoise_fn))
    np.savez_compressed(os.path.join(HERE, "c1_small_chain.npz"), **out)


def g6():
    """`relative` configs (configs/relative-small-v100-4.yaml: in_channel 9, 6-channel conditioning views,
    experiment.py:274-283): training loss + grads and one p_mean_variance, tiny net."""
    hp = dict(TINY, in_channel=9)
    vf = make_vf(hp, SCHEDULES["linear_train"], True)
    g = torch.Generator().manual_seed(51)
    B, N = 3, 3
    y_0 = torch.rand(B, 3, 16, 16, generator=g)
    y_cond = torch.rand(B, N, 6, 16, 16, generator=g)
    angle = 2 * np.pi / 24 * torch.randint(0, 24, (B, 1), generator=g).float()
    noise = torch.randn(B, 3, 16, 16, generator=g)
    vc = torch.tensor([2, 3, 1], dtype=torch.long)
    t, u = draw_t_u(vf, B, seed=52)
    loss = vf(y_cond=y_cond, view_count=vc, angle=angle, y_0=y_0, noise=noise)
    loss.backward()
    out = dict(y_0=y_0.numpy(), y_cond=y_cond.numpy(), angle=angle.numpy(), noise=noise.numpy(), view_count=vc.numpy(),
               t=t.numpy(), u=u.numpy(), loss=np.float64(loss.item()))
    out.update(grads_digest(vf.denoise_fn))
    y_t = torch.randn(B, 3, 16, 16, generator=g)
    tt = torch.tensor([1999, 700, 0], dtype=torch.long)
    with torch.no_grad():
        mean, logvar, logits, w = vf.p_mean_variance(y_t, y_cond, vc, angle, tt, clip_denoised=True)
    out.update(y_t=y_t.numpy(), pmv_t=tt.numpy(), mean=mean.numpy(), logvar=logvar.numpy(), logits=logits.numpy(),
               weights=w.numpy())
    np.savez_compressed(os.path.join(HERE, "train_relative.npz"), **out)


def g7():
    """UNet(dropout=0.1) in training mode (unet.py:207-216, Dropout in block2 of every residual block): the masks the
    reference drew are recorded by forward hooks, so the other side can replay them."""
    from torch import nn
    hp = dict(TINY, dropout=0.1)
    net = UNet(**hp)
    deterministic_fill_(net.state_dict())
    net.train()
    masks = []

    def hook(_m, inp, outp):
        masks.append((outp != 0) | (inp[0] == 0))          # kept elements (an exactly-zero input counts as kept)

    for m in net.modules():
        if isinstance(m, nn.Dropout):
            m.register_forward_hook(hook)
    g = torch.Generator().manual_seed(61)
    S = 3
    x = torch.rand(S, 6, 16, 16, generator=g).requires_grad_(True)
    angle = 2 * np.pi / 24 * torch.randint(0, 24, (S, 1), generator=g).float()
    level = torch.rand(S, 1, generator=g)
    torch.manual_seed(62)
    y = net(x, angle, level)
    gy = torch.randn(y.shape, generator=g)
    (y * gy).sum().backward()
    out = dict(x=x.detach().numpy(), angle=angle.numpy(), level=level.numpy(), y=y.detach().numpy(), gy=gy.numpy(),
               gx=x.grad.numpy(), p=np.float64(0.1), n_masks=np.array(len(masks)))
    for i, m in enumerate(masks):
        out[f"mask{i}.shape"] = np.array(m.shape)
        out[f"mask{i}.bits"] = np.packbits(m.numpy().reshape(-1))
    out.update(grads_digest(net))
    np.savez_compressed(os.path.join(HERE, "unet_tiny_dropout.npz"), **out)


def g8():
    """Checkpoint wire format (utils/checkpoint.py:31-72): the reference modules + torch Adam take one step, the file is
    written by the reference's own Checkpoint.save, then training continues for two more steps; the continued
    parameters are the expected result of `load -> two steps` on the other side."""
    from utils.checkpoint import Checkpoint
    vf = make_vf(MICRO, SCHEDULES["linear_train"], True)
    opt = torch.optim.Adam(vf.parameters(), lr=1e-4)
    ck = Checkpoint(HERE, device=torch.device("cpu"), rank=0, config=None, model=vf, optimizer=opt)
    out, B, N = {}, 3, 3
    vc = torch.tensor([3, 1, 2], dtype=torch.long)
    for step in range(3):
        y_0, y_cond, angle, noise = inputs(B, N, 16, seed=70 + step)
        t, u = draw_t_u(vf, B, seed=80 + step)
        opt.zero_grad()
        loss = vf(y_cond=y_cond, view_count=vc, angle=angle, y_0=y_0, noise=noise)
        loss.backward()
        opt.step()
        out.update({f"s{step}.y_0": y_0.numpy(), f"s{step}.y_cond": y_cond.numpy(), f"s{step}.angle": angle.numpy(),
                    f"s{step}.noise": noise.numpy(), f"s{step}.t": t.numpy(), f"s{step}.u": u.numpy(),
                    f"s{step}.loss": np.float64(loss.item())})
        if step == 0:
            quiet(ck.save, "ckpt_ref_micro.pt", it=0, t=1.5, run_id="golden", ssim=-np.inf, psnr=-np.inf)
    out["view_count"] = vc.numpy()
    for k, v in vf.state_dict().items():
        out[f"final.{k}"] = v.numpy()
    sd = opt.state_dict()
    out["final.opt.step"] = np.array([float(sd["state"][i]["step"]) for i in sorted(sd["state"])])
    np.savez_compressed(os.path.join(HERE, "ckpt_continue.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g0", "g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8"]
    for name in which:
        globals()[name]()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
