"""End-to-end parity on a real MI355X: the HIP UNet / ViewFusion against the golden vectors made
from the real reference (tests/golden) and against the CPU oracle on the same inputs.

Stated fp32 tolerances: UNet forward atol 5e-5 / rtol 1e-4; loss rel 1e-5;
gradient digests l2 rel 1e-4 (+ per-element floor 3e-5 for analytically-zero gradients).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, SMALL, TINY

pytestmark = pytest.mark.gpu
SCHED_TRAIN = dict(schedule="linear", num_timesteps=2000, linear_start=1e-6, linear_end=1e-2)
SCHED_TEST = dict(schedule="linear", num_timesteps=1000, linear_start=1e-4, linear_end=0.09)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def make_unet(hp, dev):
    from view_fusion_amd import UNet
    from view_fusion_amd.utils import deterministic_fill_
    net = UNet(**hp)
    deterministic_fill_(net.state_dict())
    return net.to(dev)


def make_vf(hp, sched, dev, weighting=True):
    from view_fusion_amd import ViewFusion
    vf = ViewFusion(make_unet(hp, dev), {"train": sched}, weighting, weighting)
    vf.set_new_noise_schedule(device=dev, phase="train")
    return vf


def check_grads(g, named_params, prefix="g."):
    from view_fusion_amd.utils import tensor_digest
    for k, p in named_params:
        ref = g[f"{prefix}{k}.stat"]
        d = tensor_digest(p.grad)
        assert abs(d["l2"] - ref[1]) <= 1e-4 * ref[1] + 3e-5 * p.numel() ** 0.5, k
        np.testing.assert_allclose(d["samples"], g[f"{prefix}{k}.samples"], rtol=2e-3,
                                   atol=2e-5 * ref[2] + 3e-5, err_msg=k)


def T(a, dev):
    return torch.tensor(a).to(dev)


def test_unet_tiny_forward_backward_vs_reference(dev):
    g = load("unet_tiny.npz")
    net = make_unet(TINY, dev)
    x = T(g["x"], dev).requires_grad_(True)
    y = net(x, T(g["angle"], dev), T(g["level"], dev))
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=1e-4, atol=5e-5)
    (y * T(g["gy"], dev)).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["gx"], rtol=1e-3, atol=3e-5)
    check_grads(g, net.named_parameters())


def test_unet_small_forward_vs_reference(dev):
    g = load("unet_small.npz")
    net = make_unet(SMALL, dev)
    with torch.no_grad():
        y = net(T(g["x"], dev), T(g["angle"], dev), T(g["level"], dev))
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("tag", ["uniform_w", "ragged_w", "ragged_mean"])
def test_train_forward_loss_and_grads_vs_reference(dev, tag):
    g = load(f"train_{tag}.npz")
    vf = make_vf(TINY, SCHED_TRAIN, dev, bool(g["weighting"]))
    loss = vf(y_cond=T(g["y_cond"], dev), view_count=torch.tensor(g["view_count"]), angle=T(g["angle"], dev),
              y_0=T(g["y_0"], dev), noise=T(g["noise"], dev), t=T(g["t"], dev), u=T(g["u"], dev))
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    loss.backward()
    check_grads(g, vf.denoise_fn.named_parameters())


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("tag,weighting", [("w", True), ("mean", False)])
def test_generate_chain_vs_reference(dev, tag, weighting, use_graph):
    g = load(f"sample_generate_{tag}.npz")
    vf = make_vf(TINY, dict(schedule="linear", num_timesteps=10, linear_start=1e-4, linear_end=0.09), dev, weighting)
    y, ret, logit_arr, weight_arr, samples = vf.generate(
        T(g["y_cond"], dev), torch.tensor(g["view_count"]), T(g["angle"], dev), y_t=T(g["y_T"], dev),
        z_seq=T(g["z_seq"], dev), use_graph=use_graph)
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), g["ret"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(samples.cpu().numpy(), g["samples"], rtol=1e-4, atol=5e-5)
    if weighting:
        np.testing.assert_allclose(logit_arr.cpu().numpy(), g["logit_arr"], rtol=1e-4, atol=5e-5)
        np.testing.assert_allclose(weight_arr.cpu().numpy(), g["weight_arr"], rtol=1e-4, atol=1e-5)
    else:
        assert logit_arr == [None] * 10 and weight_arr == [None] * 10


def test_p_mean_variance_vs_reference(dev):
    g = load("sample_pmv.npz")
    vf = make_vf(TINY, SCHED_TEST, dev, True)
    with torch.no_grad():
        mean, logvar, logits, w = vf.p_mean_variance(T(g["y_t"], dev), T(g["y_cond"], dev),
                                                     torch.tensor(g["view_count"]), T(g["angle"], dev),
                                                     T(g["t"], dev), clip_denoised=True)
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean"], rtol=1e-4, atol=5e-5)
    np.testing.assert_array_equal(logvar.cpu().numpy(), g["logvar"])
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(w.cpu().numpy(), g["weights"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("winograd", [False, True])
def test_small_train_step_vs_oracle(dev, winograd):
    """Full-size network (33.9 M params), B=2 N=2: loss and every parameter gradient vs the CPU oracle.
    winograd=True forces the fused Winograd kernels (and their one-launch weight packing) onto all
    eligible layers, as the S=96 training step uses them."""
    from oracle import unet_ref, view_fusion_ref as vfr
    from view_fusion_amd import ops
    ops.FORCE_WINOGRAD = winograd
    try:
        _small_train_step(dev)
    finally:
        ops.FORCE_WINOGRAD = False


def test_small_train_step_natural_policy(dev):
    """Same check at S = 13 ragged views with the kernel-selection policy left alone: Winograd layers with every
    tile K-split (fewer tiles than CUs), direct kernels on the 8x8 maps, concat-free decoder blocks."""
    _small_train_step(dev, B=5, N=4, vc=[3, 1, 4, 2, 3], t=[1500, 3, 700, 1999, 42])


def _small_train_step(dev, B=2, N=2, vc=(2, 1), t=(1500, 3)):
    from oracle import unet_ref, view_fusion_ref as vfr
    vf = make_vf(SMALL, SCHED_TRAIN, dev, True)
    g = torch.Generator().manual_seed(0)
    y_0, y_cond = torch.rand(B, 3, 64, 64, generator=g), torch.rand(B, N, 3, 64, 64, generator=g)
    angle = 2 * np.pi / 24 * torch.randint(0, 24, (B, 1), generator=g).float()
    noise, t, u = torch.randn(B, 3, 64, 64, generator=g), torch.tensor(list(t)), torch.rand(B, 1, generator=g)
    vc = torch.tensor(list(vc))
    loss = vf(y_cond=y_cond.to(dev), view_count=vc, angle=angle.to(dev), y_0=y_0.to(dev), noise=noise.to(dev),
              t=t.to(dev), u=u.to(dev))
    loss.backward()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in vf.denoise_fn.state_dict().items()}
    sched = vfr.schedule_buffers(vfr.beta_schedule(**SCHED_TRAIN))
    fn = lambda x, a, l: unet_ref.unet_forward(sd, SMALL, x, a, l)
    lref = vfr.train_loss(fn, sched, y_cond, vc, angle, y_0, t, u, noise, True)
    lref.backward()
    assert abs(loss.item() - lref.item()) <= 1e-5 * abs(lref.item())
    worst = 0.0
    for k, p in vf.denoise_fn.named_parameters():
        a, b = p.grad.detach().cpu().double(), sd[k].grad.double()
        err = float((a - b).norm() / b.norm().clamp_min(1e-12))
        if float(b.norm()) > 1e-4:
            worst = max(worst, err)
    assert worst < 1e-4, worst


def test_three_adam_steps_track_the_oracle(dev):
    """Multi-step training (fused Adam on the GPU) vs the CPU oracle + torch Adam with identical
    injected randomness: catches stale packed weights / state carried wrongly across steps."""
    from oracle import unet_ref, view_fusion_ref as vfr
    from view_fusion_amd import train
    vf = make_vf(TINY, SCHED_TRAIN, dev, True)
    tr = train.Trainer(vf, lr_warmup=1)
    tr.it = 0                                                   # lr = peak 1e-4 from the first step
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in vf.denoise_fn.state_dict().items()}
    opt = torch.optim.Adam(list(sd.values()), lr=1e-4)
    sched = vfr.schedule_buffers(vfr.beta_schedule(**SCHED_TRAIN))
    fn = lambda x, a, l: unet_ref.unet_forward(sd, TINY, x, a, l)
    g = torch.Generator().manual_seed(4)
    B, N = 3, 3
    vc = torch.tensor([3, 1, 2])
    losses = []
    for step in range(3):
        y_0, y_cond = torch.rand(B, 3, 16, 16, generator=g), torch.rand(B, N, 3, 16, 16, generator=g)
        angle = 2 * np.pi / 24 * torch.randint(0, 24, (B, 1), generator=g).float()
        noise, u = torch.randn(B, 3, 16, 16, generator=g), torch.rand(B, 1, generator=g)
        t = torch.randint(1, 2000, (B,), generator=g)
        batch = dict(y_0=y_0.to(dev), y_cond=y_cond.to(dev), angle=angle.to(dev), view_count=vc)
        lg = tr.step(batch, noise=noise.to(dev), t=t.to(dev), u=u.to(dev))
        opt.zero_grad()
        lc = vfr.train_loss(fn, sched, y_cond, vc, angle, y_0, t, u, noise, True)
        lc.backward()
        opt.step()
        losses.append((lg.item(), lc.item()))
    for a, b in losses:
        assert abs(a - b) <= 2e-5 * abs(b), losses
    # Adam normalises every gradient to ~+-lr per step, also the analytically-zero ones that are pure
    # round-off (e.g. biases in front of a GroupNorm with one channel per group), so single elements
    # may legitimately differ by up to steps*lr = 3e-4; the bulk must agree tightly.
    d = torch.cat([(p.cpu() - sd[k].detach()).abs().reshape(-1) for k, p in vf.denoise_fn.state_dict().items()])
    assert float(d.max()) < 3.1e-4
    assert float((d > 1e-5).float().mean()) < 0.02
    moved = max(float((p.cpu() - q).abs().max()) for (k, p), q in
                zip(vf.denoise_fn.state_dict().items(), make_unet(TINY, torch.device("cpu")).state_dict().values()))
    assert moved > 1e-4                                           # the weights really were updated


def test_psnr_and_sampler_drivers(dev):
    """§8(f) rows: PSNR kernel vs the reference formula; extrapolation (more views than trained on,
    ragged 7..23) and the autoregressive rollout (view_count 1 -> k) run through generate()."""
    from view_fusion_amd import drivers
    g = torch.Generator().manual_seed(2)
    a, b = torch.rand(5, 3, 16, 16, generator=g), torch.rand(5, 3, 16, 16, generator=g)
    ref = 20 * torch.log10(1.0 / torch.sqrt(torch.mean((a - b) ** 2, dim=(1, 2, 3))))
    got = drivers.compute_psnr(a.to(dev), b.to(dev))
    assert float((got.cpu() - ref).abs().max()) < 1e-4
    vf = make_vf(TINY, dict(schedule="linear", num_timesteps=10, linear_start=1e-4, linear_end=0.09), dev, True)
    cond = torch.rand(2, 23, 3, 16, 16, generator=g).to(dev)
    angle = torch.rand(2, 1, generator=g).to(dev)
    ret, logit_arr, weight_arr, vc = drivers.extrapolate(vf, cond, angle, max_views=6, generator=g)
    S, mx = int(vc.sum()), int(vc.max())
    assert ret.shape == (2, 11, 3, 16, 16) and logit_arr.shape == (S, 10, 3, 16, 16)
    assert weight_arr.shape == (2, 10, mx, 3, 16, 16) and 7 <= int(vc.min()) and mx <= 23
    assert float((weight_arr.sum(dim=2) - 1).abs().max()) < 1e-5 and bool(torch.isfinite(ret).all())
    orbit = drivers.autoregressive_rollout(vf, cond[:, 0], steps=4)
    assert orbit.shape == (2, 4, 3, 16, 16) and bool(torch.isfinite(orbit).all())


def test_gradient_arena_matches_plain_training(dev):
    """§8(e): the data-parallel reducer on the GPU over RCCL (a world of one rank -- the box has one GPU): with
    the gradient arena every dW/db/dgamma/dbeta is written into the communication buffer by the backward kernels
    (no gradient is copied from the second iteration on), segments are all-reduced asynchronously, and the
    parameters after three Adam steps are bit-identical to single-process training (same kernels, only the
    destination of the gradients differs).  World-size-2 semantics are covered on CPU (tests/test_ddp_gloo.py)."""
    import socket
    import torch.distributed as dist
    from view_fusion_amd import reducer, train
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    g = torch.Generator().manual_seed(9)
    B, N = 4, 3
    batches = []
    for _ in range(3):
        batches.append(dict(y_0=torch.rand(B, 3, 16, 16, generator=g).to(dev),
                            y_cond=torch.rand(B, N, 3, 16, 16, generator=g).to(dev),
                            angle=(2 * np.pi / 24 * torch.randint(0, 24, (B, 1), generator=g).float()).to(dev),
                            view_count=torch.tensor([3, 1, 2, 3]),
                            noise=torch.randn(B, 3, 16, 16, generator=g).to(dev),
                            t=torch.randint(1, 2000, (B,), generator=g).to(dev), u=torch.rand(B, 1, generator=g).to(dev)))

    def run(world):
        vf = make_vf(TINY, SCHED_TRAIN, dev, True)
        tr = train.Trainer(vf, world=world, lr_warmup=1)
        tr.it = 0
        copied = []
        for b in batches:
            b = dict(b)
            extra = {k: b.pop(k) for k in ("noise", "t", "u")}
            if tr.arena is not None:
                tr.arena.copied = 0
            tr.step(b, **extra)
            copied.append(tr.arena.copied if tr.arena is not None else None)
        return vf, tr, copied

    plain, _, _ = run(1)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        vf, tr, copied = run(2)                          # world=2 only selects the reducer; the group has one rank
        a = tr.arena
        assert a is not None and reducer.ACTIVE is a
        assert copied[1:] == [0, 0], copied              # zero-copy from the second iteration on
        assert all(p.grad.data_ptr() == a.base + 4 * a.off[i] for i, p in enumerate(a.params))
        assert len(a.seg_range) >= 2 and a.seg_range[-1][1] == a.flat.numel()
        for (k, p), q in zip(vf.state_dict().items(), plain.state_dict().values()):
            assert torch.equal(p, q), k
    finally:
        reducer.ACTIVE = None
        dist.destroy_process_group()
