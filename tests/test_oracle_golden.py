"""Pin the CPU oracle (oracle/) to the golden vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, MICRO, SMALL, TINY, SCHED_C1, c1_chain_inputs, check_digest
from oracle import unet_ref, view_fusion_ref as vfr
from view_fusion_amd.utils import deterministic_fill_, tensor_digest
from view_fusion_amd.unet import UNet

SCHED = {
    "linear_train": dict(schedule="linear", num_timesteps=2000, linear_start=1e-6, linear_end=1e-2),
    "linear_test": dict(schedule="linear", num_timesteps=1000, linear_start=1e-4, linear_end=0.09),
    "quad": dict(schedule="quad", num_timesteps=10, linear_start=1e-4, linear_end=0.09),
    "warmup10": dict(schedule="warmup10", num_timesteps=20, linear_start=1e-4, linear_end=0.09),
    "warmup50": dict(schedule="warmup50", num_timesteps=10, linear_start=1e-4, linear_end=0.09),
    "const": dict(schedule="const", num_timesteps=10, linear_start=1e-4, linear_end=0.09),
    "jsd": dict(schedule="jsd", num_timesteps=10),
    "cosine": dict(schedule="cosine", num_timesteps=10),
}


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def filled_sd(hp):
    """state_dict of the (parameter-holder) UNet with the deterministic fill."""
    net = UNet(**hp)
    sd = net.state_dict()
    deterministic_fill_(sd)
    return {k: v.clone().requires_grad_(True) for k, v in sd.items()}


@pytest.mark.parametrize("name", sorted(SCHED))
def test_schedule_buffers_exact(name):
    g = load("schedules.npz")
    with np.errstate(divide="ignore", invalid="ignore"):
        bufs = vfr.schedule_buffers(vfr.beta_schedule(**SCHED[name]))
    for k in vfr.SCHEDULE_KEYS:
        np.testing.assert_array_equal(bufs[k].numpy(), g[f"{name}.{k}"], err_msg=f"{name}.{k}")


def test_schedule_unknown_raises():
    with pytest.raises(NotImplementedError):
        vfr.beta_schedule("nope", 10)


def _check_grads(g, sd, prefix="g."):
    for k, p in sd.items():
        key = f"{prefix}{k}.stat"
        if key not in g.files:
            continue
        d = tensor_digest(p.grad)
        ref = g[key]
        # 1e-4 relative, plus a per-element absolute floor of 3e-5 for gradients that are
        # analytically ~0 (e.g. a per-channel constant added in front of a GroupNorm)
        assert abs(d["l2"] - ref[1]) <= 1e-4 * ref[1] + 3e-5 * p.numel() ** 0.5, k
        np.testing.assert_allclose(d["samples"], g[f"{prefix}{k}.samples"], rtol=2e-3,
                                   atol=2e-5 * ref[2] + 3e-5, err_msg=k)


def test_unet_tiny_forward_backward():
    g = load("unet_tiny.npz")
    sd = filled_sd(TINY)
    x = torch.tensor(g["x"], requires_grad=True)
    y = unet_ref.unet_forward(sd, TINY, x, torch.tensor(g["angle"]), torch.tensor(g["level"]))
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-4, atol=5e-5)
    (y * torch.tensor(g["gy"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["gx"], rtol=1e-3, atol=1e-5)
    _check_grads(g, sd)


def test_unet_small_forward():
    g = load("unet_small.npz")
    sd = filled_sd(SMALL)
    with torch.no_grad():
        y = unet_ref.unet_forward(sd, SMALL, torch.tensor(g["x"]), torch.tensor(g["angle"]),
                                  torch.tensor(g["level"]))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("tag", ["uniform_w", "ragged_w", "ragged_mean"])
def test_train_loss_and_grads(tag):
    g = load(f"train_{tag}.npz")
    sd = filled_sd(TINY)
    sched = vfr.schedule_buffers(vfr.beta_schedule(**SCHED["linear_train"]))
    fn = lambda x, a, l: unet_ref.unet_forward(sd, TINY, x, a, l)
    loss = vfr.train_loss(fn, sched, torch.tensor(g["y_cond"]), g["view_count"], torch.tensor(g["angle"]),
                          torch.tensor(g["y_0"]), torch.tensor(g["t"]), torch.tensor(g["u"]),
                          torch.tensor(g["noise"]), weighting=bool(g["weighting"]))
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    loss.backward()
    _check_grads(g, sd)


@pytest.mark.parametrize("tag,weighting", [("w", True), ("mean", False)])
def test_generate_chain(tag, weighting):
    g = load(f"sample_generate_{tag}.npz")
    sd = {k: v.detach() for k, v in filled_sd(TINY).items()}
    sched = vfr.schedule_buffers(vfr.beta_schedule("linear", 10, 1e-4, 0.09))
    fn = lambda x, a, l: unet_ref.unet_forward(sd, TINY, x, a, l)
    with torch.no_grad():
        y, ret, logit_arr, weight_arr, samples = vfr.generate(
            fn, sched, torch.tensor(g["y_cond"]), g["view_count"], torch.tensor(g["angle"]),
            torch.tensor(g["y_T"]), torch.tensor(g["z_seq"]), weighting=weighting)
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(ret.numpy(), g["ret"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(samples.numpy(), g["samples"], rtol=1e-4, atol=5e-5)
    assert ret.shape == (2, 11, 3, 16, 16)
    if weighting:
        assert logit_arr.shape == (3, 10, 3, 16, 16) and weight_arr.shape == (2, 10, 2, 3, 16, 16)
        np.testing.assert_allclose(logit_arr.numpy(), g["logit_arr"], rtol=1e-4, atol=5e-5)
        np.testing.assert_allclose(weight_arr.numpy(), g["weight_arr"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(weight_arr.sum(dim=2).numpy(), 1.0, atol=1e-5)
    else:
        assert logit_arr == [None] * 10 and weight_arr == [None] * 10


def test_p_mean_variance_real_schedule():
    g = load("sample_pmv.npz")
    sd = {k: v.detach() for k, v in filled_sd(TINY).items()}
    sched = vfr.schedule_buffers(vfr.beta_schedule(**SCHED["linear_test"]))
    fn = lambda x, a, l: unet_ref.unet_forward(sd, TINY, x, a, l)
    with torch.no_grad():
        mean, logvar, logits, w = vfr.p_mean_variance(fn, sched, torch.tensor(g["y_t"]), torch.tensor(g["y_cond"]),
                                                     g["view_count"], torch.tensor(g["angle"]), torch.tensor(g["t"]))
    np.testing.assert_allclose(mean.numpy(), g["mean"], rtol=1e-4, atol=5e-5)
    np.testing.assert_array_equal(logvar.numpy(), g["logvar"])
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(w.numpy(), g["weights"], rtol=1e-4, atol=1e-5)


def test_c1_small_unet_chain_and_train_step():
    """BASELINE config C1 exactly as stated (small UNet, B=2 N=2 64x64, 10 DDPM steps, CPU): the oracle reproduces
    the reference's chain with the injected y_T / z and the reference's training loss + every gradient."""
    g = load("c1_small_chain.npz")
    y_0, y_cond, angle, noise, vc, y_T, z_seq = c1_chain_inputs(g)
    sd = filled_sd(SMALL)
    sdd = {k: v.detach() for k, v in sd.items()}
    sched = vfr.schedule_buffers(vfr.beta_schedule(**SCHED_C1))
    with torch.no_grad():
        y, ret, logit_arr, weight_arr, samples = vfr.generate(
            lambda x, a, l: unet_ref.unet_forward(sdd, SMALL, x, a, l), sched, y_cond, vc, angle, y_T, z_seq)
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(samples.numpy(), g["samples"], rtol=1e-4, atol=5e-5)
    check_digest(ret, g, "ret")
    check_digest(logit_arr, g, "logit_arr")
    check_digest(weight_arr, g, "weight_arr", atol=1e-5)
    sched = vfr.schedule_buffers(vfr.beta_schedule(**SCHED["linear_train"]))
    loss = vfr.train_loss(lambda x, a, l: unet_ref.unet_forward(sd, SMALL, x, a, l), sched, y_cond, vc, angle, y_0,
                          torch.tensor(g["t"]), torch.tensor(g["u"]), noise, True)
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    loss.backward()
    _check_grads(g, sd)


def test_relative_conditioning():
    """in_channel 9 / 6-channel conditioning views (configs/relative-small-v100-4.yaml)."""
    g = load("train_relative.npz")
    hp = dict(TINY, in_channel=9)
    sd = filled_sd(hp)
    sched = vfr.schedule_buffers(vfr.beta_schedule(**SCHED["linear_train"]))
    fn = lambda x, a, l: unet_ref.unet_forward(sd, hp, x, a, l)
    loss = vfr.train_loss(fn, sched, torch.tensor(g["y_cond"]), g["view_count"], torch.tensor(g["angle"]),
                          torch.tensor(g["y_0"]), torch.tensor(g["t"]), torch.tensor(g["u"]), torch.tensor(g["noise"]))
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    loss.backward()
    _check_grads(g, sd)
    with torch.no_grad():
        mean, logvar, logits, w = vfr.p_mean_variance(fn, sched, torch.tensor(g["y_t"]), torch.tensor(g["y_cond"]),
                                                     g["view_count"], torch.tensor(g["angle"]), torch.tensor(g["pmv_t"]))
    np.testing.assert_allclose(mean.numpy(), g["mean"], rtol=1e-4, atol=5e-5)
    np.testing.assert_array_equal(logvar.numpy(), g["logvar"])
    np.testing.assert_allclose(w.numpy(), g["weights"], rtol=1e-4, atol=1e-5)


def dropout_draws(g):
    """The recorded Dropout masks of unet_tiny_dropout.npz as uniform draws: kept -> 1.0 (>= p), dropped -> 0.0 (< p)."""
    out = []
    for i in range(int(g["n_masks"])):
        shape = tuple(g[f"mask{i}.shape"])
        bits = np.unpackbits(g[f"mask{i}.bits"])[:int(np.prod(shape))]
        out.append(torch.tensor(bits.reshape(shape), dtype=torch.float32))
    return out


def test_unet_dropout_training_mode():
    g = load("unet_tiny_dropout.npz")
    hp = dict(TINY, dropout=float(g["p"]))
    sd = filled_sd(hp)
    x = torch.tensor(g["x"], requires_grad=True)
    y = unet_ref.unet_forward(sd, hp, x, torch.tensor(g["angle"]), torch.tensor(g["level"]), dropout_u=dropout_draws(g))
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-4, atol=5e-5)
    (y * torch.tensor(g["gy"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["gx"], rtol=1e-3, atol=1e-5)
    _check_grads(g, sd)


def test_checkpoint_written_by_the_reference():
    """utils/checkpoint.py wire format: the file tests/golden/ckpt_ref_micro.pt was written by the reference's own
    Checkpoint.save after one Adam step.  The product's loader reads it into the (CPU-resident) parameter-holder
    modules + torch Adam, the oracle continues training for two steps from that state and lands on the reference's
    continued parameters."""
    from view_fusion_amd import ViewFusion, drivers
    g = load("ckpt_continue.npz")
    net = UNet(**MICRO)
    vf = ViewFusion(net, {"train": SCHED["linear_train"]})
    vf.set_new_noise_schedule(device=torch.device("cpu"), phase="train")
    opt = torch.optim.Adam(vf.parameters(), lr=1e-4)
    rest = drivers.load_checkpoint(os.path.join(GOLDEN, "ckpt_ref_micro.pt"), vf, opt, device="cpu")
    assert rest["it"] == 0 and rest["t"] == 1.5 and rest["run_id"] == "golden"
    assert rest["ssim"] == -np.inf and rest["psnr"] == -np.inf
    assert all(float(st["step"]) == 1.0 for st in opt.state.values()) and len(opt.state) == len(list(vf.parameters()))
    sd = dict(net.named_parameters())
    sched = vfr.schedule_buffers(vfr.beta_schedule(**SCHED["linear_train"]))
    fn = lambda x, a, l: unet_ref.unet_forward(sd, MICRO, x, a, l)
    for step in (1, 2):
        opt.zero_grad()
        loss = vfr.train_loss(fn, sched, torch.tensor(g[f"s{step}.y_cond"]), g["view_count"],
                              torch.tensor(g[f"s{step}.angle"]), torch.tensor(g[f"s{step}.y_0"]),
                              torch.tensor(g[f"s{step}.t"]), torch.tensor(g[f"s{step}.u"]),
                              torch.tensor(g[f"s{step}.noise"]))
        assert abs(loss.item() - float(g[f"s{step}.loss"])) <= 2e-5 * abs(float(g[f"s{step}.loss"]))
        loss.backward()
        opt.step()
    # Adam normalises every gradient to ~+-lr per step, also analytically-zero ones that are pure round-off (a
    # per-channel constant in front of a GroupNorm with one channel per group), so single elements may differ by up
    # to steps*lr = 2e-4; the bulk must agree tightly.
    d = np.concatenate([np.abs(v.numpy() - g[f"final.{k}"]).reshape(-1) for k, v in vf.state_dict().items()])
    assert d.max() < 2.1e-4 and (d > 1e-5).mean() < 0.02, (d.max(), (d > 1e-5).mean())
    np.testing.assert_array_equal(np.array([float(st["step"]) for st in opt.state.values()]), g["final.opt.step"])


def test_checkpoint_saved_here_loads_in_the_reference(tmp_path):
    """The other direction: a file written by drivers.save_checkpoint is accepted by the reference's own
    Checkpoint.load into the reference's modules (skipped where /root/reference is absent, e.g. the GPU box)."""
    import sys
    if not os.path.isdir("/root/reference/utils"):
        pytest.skip("reference not present")
    from view_fusion_amd import ViewFusion, drivers
    net = UNet(**MICRO)
    deterministic_fill_(net.state_dict())
    vf = ViewFusion(net, {"train": SCHED["linear_train"]})
    vf.set_new_noise_schedule(device=torch.device("cpu"), phase="train")
    opt = torch.optim.Adam(vf.parameters(), lr=1e-4)
    for p in vf.parameters():                       # one synthetic Adam step so that the optimizer has state
        p.grad = torch.full_like(p, 1e-3)
    opt.step()
    path = str(tmp_path / "model.pt")
    drivers.save_checkpoint(path, vf, opt, it=7, t=3.25, run_id="r", ssim=0.5, psnr=20.0)
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    try:
        from model.unet import UNet as RefUNet
        from model.view_fusion import ViewFusion as RefVF
        from utils.checkpoint import Checkpoint
    finally:
        sys.path.remove("/root/reference")
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        rvf = RefVF(RefUNet(**MICRO), {"train": SCHED["linear_train"]})
        rvf.set_new_noise_schedule(device=torch.device("cpu"), phase="train")
        ropt = torch.optim.Adam(rvf.parameters(), lr=1e-4)
        rest = Checkpoint(str(tmp_path), device=torch.device("cpu"), rank=0, model=rvf, optimizer=ropt).load("model.pt")
    assert rest == dict(it=7, t=3.25, run_id="r", ssim=0.5, psnr=20.0)
    for (k, a), (k2, b) in zip(vf.state_dict().items(), rvf.state_dict().items()):
        assert k == k2 and torch.equal(a, b), k
    for a, b in zip(opt.state.values(), ropt.state.values()):
        assert torch.equal(a["exp_avg"], b["exp_avg"]) and float(a["step"]) == float(b["step"])
