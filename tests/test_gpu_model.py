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
    ops.st.FORCE_WINOGRAD = winograd
    try:
        _small_train_step(dev)
    finally:
        ops.st.FORCE_WINOGRAD = False


def test_small_train_step_natural_policy(dev):
    """Same check at S = 13 ragged views with the kernel-selection policy left alone: Winograd layers with every
    tile K-split (fewer tiles than CUs), direct kernels on the 8x8 maps, concat-free decoder blocks."""
    _small_train_step(dev, B=5, N=4, vc=[3, 1, 4, 2, 3], t=[1500, 3, 700, 1999, 42])


def test_weight_gradient_slab_sums_are_deferred_into_one_launch(dev):
    """Round 5: every Winograd weight gradient runs only its main kernel (vf_wino_wgrad_main, a workspace of its own) and
    ONE vf_wino44_reduce_multi launch at the end of the backward pass fills all dW / db -- bitwise what the per-layer
    launches give (same slabs, same order).  A pass that ACCUMULATES into existing .grad tensors must not defer (the
    destination would be read before it is filled)."""
    from view_fusion_amd import ops
    vf = make_vf(SMALL, SCHED_TRAIN, dev, True)
    g = torch.Generator().manual_seed(5)
    B, N = 3, 2
    kw = dict(y_cond=torch.rand(B, N, 3, 64, 64, generator=g).to(dev), view_count=torch.tensor([2, 2, 1]),
              angle=torch.rand(B, 1, generator=g).to(dev) * 6, y_0=torch.rand(B, 3, 64, 64, generator=g).to(dev),
              noise=torch.randn(B, 3, 64, 64, generator=g).to(dev), t=torch.tensor([1500, 3, 77]).to(dev),
              u=torch.rand(B, 1, generator=g).to(dev))
    params = [p for p in vf.denoise_fn.parameters()]

    def run(defer, zero=True):
        if zero:
            for p in params:
                p.grad = None
        ops.st.WRED_DEFER, ops.st.KERNEL_LOG = defer, []
        try:
            vf(**kw).backward()
            torch.cuda.synchronize()
            return [e[5] for e in ops.st.KERNEL_LOG], [p.grad.detach().clone() for p in params]
        finally:
            ops.st.WRED_DEFER, ops.st.KERNEL_LOG = True, None

    names, g_def = run(True)
    n_main = names.count("vf_wino_wgrad_main")
    assert n_main >= 20 and names.count("vf_wino44_reduce_multi") == 1 and "vf_wino_wgrad" not in names, \
        (n_main, names.count("vf_wino44_reduce_multi"), names.count("vf_wino_wgrad"))
    # round 6: the direct / 1x1 weight gradients' slab sums ride in the same launch
    n_gen = names.count("vf_conv_wgrad_main") + names.count("vf_conv1x1_cat_wgrad_main")
    assert n_gen >= 20 and "vf_conv_wgrad" not in names and "vf_conv1x1_cat_wgrad" not in names, n_gen
    names, g_now = run(False)
    assert names.count("vf_wino_wgrad") == n_main and "vf_wino_wgrad_main" not in names
    assert names.count("vf_conv_wgrad") + names.count("vf_conv1x1_cat_wgrad") == n_gen and "vf_conv_wgrad_main" not in names
    for a, b in zip(g_def, g_now):
        assert torch.equal(a, b)
    names, g_acc = run(True, zero=False)                  # second pass on top of the existing gradients
    assert "vf_wino_wgrad_main" not in names and names.count("vf_wino_wgrad") == n_main
    assert "vf_conv_wgrad_main" not in names and "vf_conv1x1_cat_wgrad_main" not in names
    for a, b in zip(g_acc, g_now):
        assert float((a - 2 * b).abs().max()) <= 1e-6 * float(b.abs().max()) + 1e-12


def test_deferred_slab_sums_share_one_arena_across_passes_and_graphs(dev):
    """Round 6: the deferred layers' slab workspaces are slices of ONE per-device arena refilled from its
    start by every backward pass -- eager or replayed, whatever the geometry -- and a pending entry keeps neither the
    layer input nor dY.  Three captured geometries must not add a slab workspace each to the graphs' memory pool."""
    from view_fusion_amd import ops, train
    vf = make_vf(SMALL, SCHED_TRAIN, dev, True)
    tr = train.Trainer(vf, graph=True, lr_warmup=1)
    g = torch.Generator().manual_seed(3)

    def batch(B, N=2):
        return dict(y_0=torch.rand(B, 3, 64, 64, generator=g).to(dev), y_cond=torch.rand(B, N, 3, 64, 64, generator=g).to(dev),
                    angle=(torch.rand(B, 1, generator=g) * 6).to(dev), view_count=torch.full((B,), N))

    seen_entries = []
    orig = ops.deferred._flush_wred

    def spy():
        seen_entries.extend(ops.st._PENDING_WRED)
        orig()
    ops.deferred._flush_wred = spy
    try:
        b3 = batch(3)
        for _ in range(train.Trainer.GRAPH_AFTER + 2):      # eager sightings, capture, one replay
            tr.step(b3)
        torch.cuda.synchronize()
        assert tr.graph_steps >= 1
        arena0, mem0 = ops.wred_arena_bytes(dev), torch.cuda.memory_allocated(dev)
        assert arena0 > 0
        for B in (2, 1):                                     # two smaller geometries: the arena already fits them
            bb = batch(B)
            for _ in range(train.Trainer.GRAPH_AFTER + 2):
                tr.step(bb)
        torch.cuda.synchronize()
        assert len(tr._graphs) == 3 and all(e.graph is not None for e in tr._graphs.values())
        assert ops.wred_arena_bytes(dev) == arena0           # no second arena, no per-graph workspaces
        grown = torch.cuda.memory_allocated(dev) - mem0
        # each further graph owns its gradients (136 MB) + activations of its (smaller) geometry; a private set of slab
        # workspaces would add ~0.25 GB per graph at these sizes on top
        assert grown < 2.0e9, grown
    finally:
        ops.deferred._flush_wred = orig
    assert len(seen_entries) >= 20
    for row, nblk, keep in seen_entries:
        assert len(keep) == 4 and keep[0]._base is not None and keep[0].dim() == 1      # (ws = a slice of the arena, dw, db, db2)
        assert keep[1].dim() == 4 and all(t is None or t.dim() == 1 for t in keep[2:])     # no layer input, no dY


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
    parameters after eight Adam steps are bit-identical to single-process training (same kernels, only the
    destination of the gradients differs) -- launched eagerly AND with the whole iteration, the six segment
    all-reduces on RCCL's stream included, replayed as one HIP graph (the launch path of bench.py --gpus N).
    World-size-2 semantics are covered on CPU (tests/test_ddp_gloo.py) and by tests/test_gpu_two_rank.py."""
    import socket
    import torch.distributed as dist
    from view_fusion_amd import reducer, train
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    g = torch.Generator().manual_seed(9)
    B, N, STEPS = 4, 3, 8
    batches = []
    for _ in range(STEPS):
        batches.append(dict(y_0=torch.rand(B, 3, 16, 16, generator=g).to(dev),
                            y_cond=torch.rand(B, N, 3, 16, 16, generator=g).to(dev),
                            angle=(2 * np.pi / 24 * torch.randint(0, 24, (B, 1), generator=g).float()).to(dev),
                            view_count=torch.tensor([3, 1, 2, 3]),
                            noise=torch.randn(B, 3, 16, 16, generator=g).to(dev),
                            t=torch.randint(1, 2000, (B,), generator=g).to(dev), u=torch.rand(B, 1, generator=g).to(dev)))

    def run(world, graph):
        vf = make_vf(TINY, SCHED_TRAIN, dev, True)
        tr = train.Trainer(vf, world=world, lr_warmup=1, graph=graph)
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

    plain, _, _ = run(1, False)
    train.init_rccl_group(0, rank=0, world_size=1)            # the harness' own group setup (high-priority RCCL streams)
    try:
        # eager | captured (collectives inside the graph: the default for a group of one) | split on RCCL (what a real
        # multi-rank run defaults to: forward + backward replayed, collectives + Adam eager) | a capture that fails in
        # captured mode: the failure flag rides the all-reduce, the agreement point steps down to split, which captures
        # | the hand-written exchange (VF_REDUCER=xgmi) over the same RCCL-initialised group of one: its own rank is the
        # only "peer", the fused kernels average over one arena and apply Adam -- same parameters, bit for bit
        for graph, env, mode in ((False, {}, "eager"), (True, {}, "captured"),
                                 (True, {"VF_CAPTURE_COLLECTIVES": "0"}, "split"),
                                 (True, {"inject": "0:captured"}, "split"),
                                 (True, {"VF_REDUCER": "xgmi"}, "eager")):
            env = dict(env)
            train.Trainer.inject_capture_failure = env.pop("inject", None)
            injected = train.Trainer.inject_capture_failure is not None
            os.environ.update(env)
            try:
                vf, tr, copied = run(2, graph)           # world=2 only selects the reducer; the group has one rank
            finally:
                train.Trainer.inject_capture_failure = None
                for k in env:
                    os.environ.pop(k)
            a = tr.arena
            assert a is not None and reducer.ACTIVE is a and tr.mode == mode, (tr.mode, mode)
            assert copied[1:] == [0] * (STEPS - 1), copied   # zero-copy from the second iteration on (capture included)
            xg = env.get("VF_REDUCER") == "xgmi"
            if xg:                                       # p.grad = the averaged-gradient buffer, same layout as the arena
                assert all(p.grad.data_ptr() == a.gavg.data_ptr() + 4 * a.off[i] for i, p in enumerate(a.params))
            else:
                assert all(p.grad.data_ptr() == a.base + 4 * a.off[i] for i, p in enumerate(a.params))
            assert len(a.seg_range) >= 2 and a.seg_range[-1][1] == a.flat.numel()
            if xg:
                assert tr.graph_steps == 0 and tr.dist_info()["reducer"] == "xgmi"
            elif injected:                                 # it = 4 fails, read at it = 6, sightings 6-7, replay at it = 8
                assert tr.demotions == 1 and tr.graph_steps == 1, (tr.demotions, tr.graph_steps)
            elif graph:                                  # iterations 0-2 eager (layout, two sightings), then replays
                assert a.capturable == (mode == "captured")
                assert tr.graph_steps == STEPS - 1 - train.Trainer.GRAPH_AFTER, tr.graph_steps
            else:
                assert tr.graph_steps == 0
            for (k, p), q in zip(vf.state_dict().items(), plain.state_dict().values()):
                assert torch.equal(p, q), (mode, env, k)
            reducer.ACTIVE = None
    finally:
        reducer.ACTIVE = None
        dist.destroy_process_group()


# ================================================================================================
# Round 2: parity on the REAL workloads (BASELINE configs C1 / C2 / C4 / C5) and the remaining
# boundary branches (relative conditioning, dropout, checkpoints, sampler drivers).
# ================================================================================================
from conftest import MICRO, SCHED_C1, c1_chain_inputs, check_digest   # noqa: E402


def cpu_sd(module):
    return {k: v.detach().cpu().clone().requires_grad_(True) for k, v in module.state_dict().items()}


def oracle_train_chunked(sd, hp, sched, y_cond, vc, angle, y_0, t, u, noise, weighting=True, chunk=2):
    """Oracle loss + gradients (accumulated into sd[k].grad) of a LARGE batch, evaluated a few samples at a time:
    samples are independent through the UNet (GroupNorm / attention are per view) and the loss is the mean of the
    per-sample MSEs, so loss = sum_chunks loss_chunk * B_chunk / B.  Keeps the CPU autograd tape small."""
    from oracle import unet_ref, view_fusion_ref as vfr
    B = y_0.shape[0]
    fn = lambda x, a, l: unet_ref.unet_forward(sd, hp, x, a, l)
    total = 0.0
    for lo in range(0, B, chunk):
        sl = slice(lo, min(B, lo + chunk))
        part = vfr.train_loss(fn, sched, y_cond[sl], vc[sl], angle[sl], y_0[sl], t[sl], u[sl], noise[sl], weighting)
        part = part * ((sl.stop - sl.start) / B)
        part.backward()
        total += float(part.item())
    return total


@pytest.mark.parametrize("B,N,ragged", [(16, 6, False), (8, 6, False), (16, 6, True)])
def test_full_size_train_step_vs_oracle(dev, B, N, ragged):
    """BASELINE C2 (B=16 N=6, S=96) and C4 (B=8 N=6, S=48) with the NATURAL kernel policy -- Winograd tail plans at
    288/144 tiles, attn_fwd_kh vs attn_fwd_split, wgrad thresholds, concat-free decoder -- plus a ragged batch as the
    reference's training loop draws it (experiment.py:277-279): loss rel 1e-5, every parameter gradient rel-L2 1e-4."""
    from oracle import view_fusion_ref as vfr
    vf = make_vf(SMALL, SCHED_TRAIN, dev, True)
    g = torch.Generator().manual_seed(100 + B)
    y_0, y_cond = torch.rand(B, 3, 64, 64, generator=g), torch.rand(B, N, 3, 64, 64, generator=g)
    angle = 2 * np.pi / 24 * torch.randint(0, 24, (B, 1), generator=g).float()
    noise, u = torch.randn(B, 3, 64, 64, generator=g), torch.rand(B, 1, generator=g)
    t = torch.randint(1, 2000, (B,), generator=g)
    vc = torch.randint(1, N + 1, (B,), generator=g) if ragged else torch.full((B,), N)
    loss = vf(y_cond=y_cond.to(dev), view_count=vc, angle=angle.to(dev), y_0=y_0.to(dev), noise=noise.to(dev),
              t=t.to(dev), u=u.to(dev))
    loss.backward()
    import oracle_pool
    sd0 = {k: v.detach().cpu() for k, v in vf.denoise_fn.state_dict().items()}
    try:                        # the oracle, sample-parallel (tests/oracle_pool.py: samples are independent)
        lref, gref = oracle_pool.train(sd0, SMALL, SCHED_TRAIN, y_cond, vc, angle, y_0, t, u, noise)
    finally:
        oracle_pool.close()
    assert abs(loss.item() - lref) <= 1e-5 * abs(lref), (loss.item(), lref)
    worst, wk = 0.0, None
    for k, p in vf.denoise_fn.named_parameters():
        a, b = p.grad.detach().cpu().double(), gref[k].double()
        if float(b.norm()) > 1e-4:
            err = float((a - b).norm() / b.norm())
            if err > worst:
                worst, wk = err, k
    assert worst < 1e-4, (worst, wk)
    if B != 16 or ragged:
        return
    # ... and the path bench.py TIMES: the same iteration replayed from the Trainer's HIP graph (learning rate 0, so
    # that every iteration starts from the same parameters; the third one is a replay) against the same oracle values
    from view_fusion_amd import train
    vf.zero_grad(set_to_none=True)
    tr = train.Trainer(vf, graph=True, lr_warmup=1)
    tr.sched.peak_lr = 0.0
    bt = dict(y_0=y_0.to(dev), y_cond=y_cond.to(dev), angle=angle.to(dev), view_count=vc)
    draws = dict(t=t.to(dev), u=u.to(dev), noise=noise.to(dev))
    for _ in range(train.Trainer.GRAPH_AFTER + 1):
        lg = tr.step(bt, **draws)
    assert tr.graph_steps == 1
    assert abs(lg.item() - lref) <= 1e-5 * abs(lref), (lg.item(), lref)
    worst, wk = 0.0, None
    for k, p in vf.denoise_fn.named_parameters():
        a, b = p.grad.detach().cpu().double(), gref[k].double()
        if float(b.norm()) > 1e-4:
            err = float((a - b).norm() / b.norm())
            if err > worst:
                worst, wk = err, k
    assert worst < 1e-4, ("graph replay", worst, wk)


@pytest.mark.parametrize("use_graph", [True, False])
def test_c1_small_unet_chain_vs_reference(dev, use_graph):
    """BASELINE C1 as stated (small UNet, B=2 N=2 64x64, 10 DDPM steps) against the vectors of the real reference."""
    g = load("c1_small_chain.npz")
    _, y_cond, angle, _, vc, y_T, z_seq = c1_chain_inputs(g)
    vf = make_vf(SMALL, SCHED_C1, dev, True)
    y, ret, logit_arr, weight_arr, samples = vf.generate(y_cond.to(dev), vc, angle.to(dev), y_t=y_T.to(dev),
                                                         z_seq=z_seq.to(dev), use_graph=use_graph)
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(samples.cpu().numpy(), g["samples"], rtol=1e-4, atol=5e-5)
    check_digest(ret.cpu(), g, "ret")
    check_digest(logit_arr.cpu(), g, "logit_arr")
    check_digest(weight_arr.cpu(), g, "weight_arr", atol=1e-5)


def test_c1_small_unet_train_step_vs_reference(dev):
    g = load("c1_small_chain.npz")
    y_0, y_cond, angle, noise, vc, _, _ = c1_chain_inputs(g)
    vf = make_vf(SMALL, SCHED_TRAIN, dev, True)
    loss = vf(y_cond=y_cond.to(dev), view_count=vc, angle=angle.to(dev), y_0=y_0.to(dev), noise=noise.to(dev),
              t=T(g["t"], dev), u=T(g["u"], dev))
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    loss.backward()
    check_grads(g, vf.denoise_fn.named_parameters())


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("N", [1, 6, 12])
def test_small_unet_sampler_vs_oracle(dev, N, use_graph):
    """BASELINE C5 shapes: small UNet 64x64, B=1, N in {1,6,12} conditioning views, 12 reverse steps of the T=1000
    schedule's kernels (graph capture with split-K workspaces + key-split attention, and eager), injected y_T / z."""
    from oracle import unet_ref, view_fusion_ref as vfr
    sched_kw = dict(schedule="linear", num_timesteps=12, linear_start=1e-4, linear_end=0.09)
    vf = make_vf(SMALL, sched_kw, dev, True)
    g = torch.Generator().manual_seed(200 + N)
    y_cond = torch.rand(1, N, 3, 64, 64, generator=g)
    angle = 2 * np.pi / 24 * torch.randint(0, 24, (1, 1), generator=g).float()
    y_T = torch.randn(1, 3, 64, 64, generator=g)
    z_seq = torch.randn(12, 1, 3, 64, 64, generator=g)
    vc = torch.tensor([N])
    y, ret, logit_arr, weight_arr, samples = vf.generate(y_cond.to(dev), vc, angle.to(dev), y_t=y_T.to(dev),
                                                         z_seq=z_seq.to(dev), use_graph=use_graph)
    sd = {k: v.detach().cpu() for k, v in vf.denoise_fn.state_dict().items()}
    sched = vfr.schedule_buffers(vfr.beta_schedule(**sched_kw))
    with torch.no_grad():
        yr, retr, lr, wr, _ = vfr.generate(lambda x, a, l: unet_ref.unet_forward(sd, SMALL, x, a, l), sched, y_cond, vc,
                                           angle, y_T, z_seq)
    np.testing.assert_allclose(y.cpu().numpy(), yr.numpy(), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), retr.numpy(), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(logit_arr.cpu().numpy(), lr.numpy(), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(weight_arr.cpu().numpy(), wr.numpy(), rtol=1e-4, atol=1e-5)


def _kernel_names(fn):
    """C-ABI entry points a call goes through (ops.st.KERNEL_LOG), in launch order."""
    from view_fusion_amd import ops
    ops.st.KERNEL_LOG = []
    try:
        fn()
        torch.cuda.synchronize()
        return [e[5] for e in ops.st.KERNEL_LOG]
    finally:
        ops.st.KERNEL_LOG = None


def test_small_unet_sampler_b16_vs_oracle(dev):
    """The path bench.py times as `sampler.B16_N6` (reference view_fusion.py:179-214 at B=16, N=6: S = 96 stacked views,
    eager -- S > 16 -- under the no-grad kernel policy: F(4x4) + nested Winograd forward, conv -> GroupNorm un-fused,
    attn_fwd_kh without the P write): three reverse steps of generate() against the oracle, tolerances as the B=1
    sampler tests; and the two Winograd forward kernels really are what ran."""
    import oracle_pool
    sched_kw = dict(schedule="linear", num_timesteps=3, linear_start=1e-4, linear_end=0.09)
    vf = make_vf(SMALL, sched_kw, dev, True)
    B, N = 16, 6
    g = torch.Generator().manual_seed(216)
    y_cond = torch.rand(B, N, 3, 64, 64, generator=g)
    angle = 2 * np.pi / 24 * torch.randint(0, 24, (B, 1), generator=g).float()
    y_T = torch.randn(B, 3, 64, 64, generator=g)
    z_seq = torch.randn(3, B, 3, 64, 64, generator=g)
    vc = torch.full((B,), N)
    args = (y_cond.to(dev), vc, angle.to(dev))
    y, ret, logit_arr, weight_arr, samples = vf.generate(*args, y_t=y_T.to(dev), z_seq=z_seq.to(dev), sample_num=2)
    names = _kernel_names(lambda: vf.p_sample(y_T.to(dev), *args, torch.full((B,), 2, device=dev), z=z_seq[2].to(dev)))
    assert "vf_wino44_conv_fwd" in names and "vf_wino_conv_fwd" in names, sorted(set(names))
    sd = {k: v.detach().cpu() for k, v in vf.denoise_fn.state_dict().items()}
    try:
        yr, retr, lr, wr, _ = oracle_pool.generate(sd, SMALL, sched_kw, y_cond, vc, angle, y_T, z_seq, sample_num=2)
    finally:
        oracle_pool.close()
    np.testing.assert_allclose(y.cpu().numpy(), yr.numpy(), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), retr.numpy(), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(logit_arr.cpu().numpy(), lr.numpy(), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(weight_arr.cpu().numpy(), wr.numpy(), rtol=1e-4, atol=1e-5)


def test_long_chain_through_winograd_kernels(dev):
    """A long reverse chain THROUGH the Winograd kernels: small UNet, B=8 N=6 (S = 48: F(4x4) on the 64x64 / 32x32 maps,
    nested kernel below), the last 100 steps (t = 99 .. 0) of the T=1000 sampler schedule with injected z, every step's
    output fed to the next.  Stated max-abs 1e-3; the measured value is printed and asserted with 10x
    margin.  One forward of the F(4x4) kernel alone carries 1e-5 (profiles/r04_parity_margin.txt): this is the test
    that shows it does not accumulate."""
    import oracle_pool
    vf = make_vf(SMALL, SCHED_TEST, dev, True)
    B, N, steps = 8, 6, 100
    g = torch.Generator().manual_seed(301)
    y_cond = torch.rand(B, N, 3, 64, 64, generator=g)
    angle = 2 * np.pi / 24 * torch.randint(0, 24, (B, 1), generator=g).float()
    # start where the chain would be at t = 99: a lightly noised image
    y_start = (0.8 * (2 * torch.rand(B, 3, 64, 64, generator=g) - 1) + 0.6 * torch.randn(B, 3, 64, 64, generator=g))
    z_seq = torch.randn(steps, B, 3, 64, 64, generator=g)
    vc = torch.full((B,), N)
    args = (y_cond.to(dev), vc, angle.to(dev))
    names = _kernel_names(lambda: vf.p_sample(y_start.to(dev), *args, torch.full((B,), 99, device=dev),
                                              z=z_seq[0].to(dev)))
    assert "vf_wino44_conv_fwd" in names and "vf_wino_conv_fwd" in names, sorted(set(names))
    y, kept = y_start.to(dev), []
    for n, i in enumerate(range(steps - 1, -1, -1)):
        y, _, w = vf.p_sample(y, *args, torch.full((B,), i, device=dev), z=z_seq[n].to(dev))
        if (n + 1) % 10 == 0:
            kept.append(y)
    sd = {k: v.detach().cpu() for k, v in vf.denoise_fn.state_dict().items()}
    try:
        yr, keptr, wr = oracle_pool.chain(sd, SMALL, SCHED_TEST, y_cond, vc, angle, y_start, z_seq, steps - 1, 0,
                                          keep_every=10)
    finally:
        oracle_pool.close()
    err = float((y.cpu() - yr).abs().max())
    err_mid = float((torch.stack(kept).cpu() - keptr).abs().max())
    err_w = float((w.cpu() - wr).abs().max())
    print(f"100-step chain through the Winograd kernels (S=48): final max-abs {err:.3e}, over every 10th step "
          f"{err_mid:.3e}, weights {err_w:.3e}")
    assert err < 1e-3 and err_mid < 1e-3          # the stated long-chain tolerance
    assert err < 1e-4 and err_mid < 1e-4, (err, err_mid)


def test_extrapolate_real_unet_n23(dev):
    """SURVEY 8(f2) on the REAL UNet: drivers.extrapolate (experiment.py:472-488) with up to 23 conditioning views,
    B=2 (S = 40 stacked views, ragged), ten reverse steps, against the oracle."""
    import oracle_pool
    from view_fusion_amd import drivers
    vf = make_vf(SMALL, SCHED_C1, dev, True)
    g = torch.Generator().manual_seed(423)
    cond, angle = torch.rand(2, 23, 3, 64, 64, generator=g), torch.rand(2, 1, generator=g) * 6
    vc = torch.tensor([23, 17])
    y_T, z_seq = torch.randn(2, 3, 64, 64, generator=g), torch.randn(10, 2, 3, 64, 64, generator=g)
    ret, logit_arr, weight_arr, _ = drivers.extrapolate(vf, cond.to(dev), angle.to(dev), view_count=vc,
                                                        y_t=y_T.to(dev), z_seq=z_seq.to(dev))
    sd = {k: v.detach().cpu() for k, v in vf.denoise_fn.state_dict().items()}
    try:
        _, retr, lr, wr, _ = oracle_pool.generate(sd, SMALL, SCHED_C1, cond, vc, angle, y_T, z_seq)
    finally:
        oracle_pool.close()
    assert weight_arr.shape == (2, 10, 23, 3, 64, 64)      # T = 10, sample_num = 8: every step is stashed
    np.testing.assert_allclose(ret.cpu().numpy(), retr.clamp(0, 1).numpy(), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(logit_arr.cpu().numpy(), lr.numpy(), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(weight_arr.cpu().numpy(), wr.numpy(), rtol=1e-4, atol=1e-5)


def test_long_chain_T1000_tolerance(dev):
    """SURVEY 8c proposed max-abs 1e-3 for a T=1000 injected-noise chain; measured here on the tiny net with the real
    sampler schedule (graph replay): the chain contracts errors (clamp + posterior mean), it does not amplify them."""
    from oracle import unet_ref, view_fusion_ref as vfr
    vf = make_vf(TINY, SCHED_TEST, dev, True)
    g = torch.Generator().manual_seed(300)
    B, N = 2, 3
    y_cond = torch.rand(B, N, 3, 16, 16, generator=g)
    angle = 2 * np.pi / 24 * torch.randint(0, 24, (B, 1), generator=g).float()
    y_T = torch.randn(B, 3, 16, 16, generator=g)
    z_seq = torch.randn(1000, B, 3, 16, 16, generator=g)
    vc = torch.tensor([3, 2])
    y, ret, *_ = vf.generate(y_cond.to(dev), vc, angle.to(dev), y_t=y_T.to(dev), z_seq=z_seq.to(dev), use_graph=True)
    import oracle_pool
    sd = {k: v.detach().cpu() for k, v in vf.denoise_fn.state_dict().items()}
    try:                                            # one worker per sample (the chain is 1000 sequential tiny forwards)
        yr, retr, *_ = oracle_pool.generate(sd, TINY, SCHED_TEST, y_cond, vc, angle, y_T, z_seq)
    finally:
        oracle_pool.close()
    err = float((y.cpu() - yr).abs().max())
    err_mid = float((ret.cpu() - retr).abs().max())
    print(f"T=1000 chain: final max-abs {err:.3e}, over all stashed steps {err_mid:.3e}")
    assert err < 1e-3 and err_mid < 1e-3          # the stated long-chain tolerance
    assert err < 1e-4, err                          # what was actually measured leaves 10x margin


def test_weights_are_fresh_after_training(dev):
    """sample -> Trainer steps (FusedAdam writes through raw pointers) -> sample: the no-grad forwards must see the
    UPDATED weights in every pack format (3x3 layers are Winograd at the training S, direct at the sampler's S).
    Compared with a fresh model loaded from the trained state_dict."""
    from view_fusion_amd import train
    vf = make_vf(SMALL, SCHED_TRAIN, dev, True)
    tr = train.Trainer(vf, lr_warmup=1)
    tr.it = 0
    tr.sched.peak_lr = 2e-3                           # large steps: stale weights would be far outside tolerance
    b = train.synthetic_batch(4, 4, 64, dev, seed=3)
    g = torch.Generator().manual_seed(7)
    y_t, z = torch.randn(1, 3, 64, 64, generator=g).to(dev), torch.randn(1, 3, 64, 64, generator=g).to(dev)
    cond, ang, t = b["y_cond"][:1, :2].contiguous(), b["angle"][:1], torch.tensor([500], device=dev)

    def sample(model):
        with torch.no_grad():
            y, _, w = model.p_sample(y_t, cond, torch.tensor([2]), ang, t, z=z)
        return y.clone(), w.clone()

    before = sample(vf)
    for _ in range(2):
        tr.step(b)
    mid = sample(vf)
    tr.step(b)
    after = sample(vf)
    fresh = make_vf(SMALL, SCHED_TRAIN, dev, True)
    fresh.load_state_dict(vf.state_dict())
    want = sample(fresh)
    assert float((after[0] - want[0]).abs().max()) < 1e-5 and float((after[1] - want[1]).abs().max()) < 1e-5
    assert float((after[0] - mid[0]).abs().max()) > 1e-4 and float((mid[0] - before[0]).abs().max()) > 1e-4


def test_relative_conditioning_vs_reference(dev):
    """configs/relative-small-v100-4.yaml: in_channel 9, 6-channel conditioning views (experiment.py:274-283)."""
    g = load("train_relative.npz")
    vf = make_vf(dict(TINY, in_channel=9), SCHED_TRAIN, dev, True)
    vc = torch.tensor(g["view_count"])
    loss = vf(y_cond=T(g["y_cond"], dev), view_count=vc, angle=T(g["angle"], dev), y_0=T(g["y_0"], dev),
              noise=T(g["noise"], dev), t=T(g["t"], dev), u=T(g["u"], dev))
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    loss.backward()
    check_grads(g, vf.denoise_fn.named_parameters())
    with torch.no_grad():
        mean, logvar, logits, w = vf.p_mean_variance(T(g["y_t"], dev), T(g["y_cond"], dev), vc.to(dev),   # device tensor
                                                     T(g["angle"], dev), T(g["pmv_t"], dev), clip_denoised=True)
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean"], rtol=1e-4, atol=5e-5)
    np.testing.assert_array_equal(logvar.cpu().numpy(), g["logvar"])
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(w.cpu().numpy(), g["weights"], rtol=1e-4, atol=1e-5)


def test_unet_dropout_vs_reference(dev):
    """UNet(dropout=0.1): training mode replays the Dropout masks the reference drew; eval mode is the identity."""
    from test_oracle_golden import dropout_draws
    g = load("unet_tiny_dropout.npz")
    hp = dict(TINY, dropout=float(g["p"]))
    net = make_unet(hp, dev).train()
    x = T(g["x"], dev).requires_grad_(True)
    y = net(x, T(g["angle"], dev), T(g["level"], dev), dropout_u=[u.to(dev) for u in dropout_draws(g)])
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=1e-4, atol=5e-5)
    (y * T(g["gy"], dev)).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["gx"], rtol=1e-3, atol=3e-5)
    check_grads(g, net.named_parameters())
    net.eval()
    plain = make_unet(TINY, dev)
    with torch.no_grad():
        a = net(x.detach(), T(g["angle"], dev), T(g["level"], dev))
        b = plain(x.detach(), T(g["angle"], dev), T(g["level"], dev))
    assert torch.equal(a, b)
    net.train()                                          # default draws: about p of block2's activations dropped
    with torch.no_grad():
        c = net(x.detach(), T(g["angle"], dev), T(g["level"], dev))
    assert float((c - a).abs().max()) > 1e-3


def test_checkpoint_from_the_reference_on_the_gpu(dev, tmp_path):
    """the file written by the reference's Checkpoint.save (model + torch Adam state after one step) is
    loaded into the HIP model + FusedAdam, two more steps land on the reference's continued parameters, and the file
    saved afterwards has the reference's layout (torch.optim.Adam loads its optimizer state)."""
    from view_fusion_amd import drivers
    from view_fusion_amd.optim import FusedAdam
    g = load("ckpt_continue.npz")
    vf = make_vf(MICRO, SCHED_TRAIN, dev, True)
    opt = FusedAdam(vf.parameters(), lr=1e-4)
    rest = drivers.load_checkpoint(os.path.join(GOLDEN, "ckpt_ref_micro.pt"), vf, opt, device=dev)
    assert rest["it"] == 0 and rest["run_id"] == "golden"
    vc = torch.tensor(g["view_count"])
    for step in (1, 2):
        opt.zero_grad()
        loss = vf(y_cond=T(g[f"s{step}.y_cond"], dev), view_count=vc, angle=T(g[f"s{step}.angle"], dev),
                  y_0=T(g[f"s{step}.y_0"], dev), noise=T(g[f"s{step}.noise"], dev), t=T(g[f"s{step}.t"], dev),
                  u=T(g[f"s{step}.u"], dev))
        assert abs(loss.item() - float(g[f"s{step}.loss"])) <= 2e-5 * abs(float(g[f"s{step}.loss"]))
        loss.backward()
        opt.step()
    d = np.concatenate([np.abs(v.cpu().numpy() - g[f"final.{k}"]).reshape(-1) for k, v in vf.state_dict().items()])
    assert d.max() < 2.1e-4 and (d > 1e-5).mean() < 0.02, (d.max(), (d > 1e-5).mean())
    path = str(tmp_path / "model.pt")
    drivers.save_checkpoint(path, vf, opt, it=2, t=1.5, run_id="golden", ssim=-np.inf, psnr=-np.inf)
    mine = torch.load(path, map_location="cpu", weights_only=False)
    ref = torch.load(os.path.join(GOLDEN, "ckpt_ref_micro.pt"), map_location="cpu", weights_only=False)
    assert list(mine.keys()) == list(ref.keys()) or set(mine) == set(ref)
    assert list(mine["model"].keys()) == list(ref["model"].keys())
    assert mine["optimizer"]["param_groups"][0]["params"] == ref["optimizer"]["param_groups"][0]["params"]
    for i, st in ref["optimizer"]["state"].items():
        ms = mine["optimizer"]["state"][i]
        assert set(ms) == set(st) and ms["exp_avg"].shape == st["exp_avg"].shape and float(ms["step"]) == 3.0
    cpu_params = [torch.nn.Parameter(v.clone()) for k, v in mine["model"].items() if k.startswith("denoise_fn.")]
    torch.optim.Adam(cpu_params, lr=1e-4).load_state_dict(mine["optimizer"])      # the reference's optimizer class


def test_sampler_drivers_vs_oracle(dev):
    """the three sampler drivers with their real call shapes -- extrapolation (ragged 7..23 views), the
    autoregressive rollout (view_count 1 -> 3) and the weight-animation call (B=24 targets x N=6 views) -- compared with
    the oracle through the injected-randomness hooks of forward(generate=True)."""
    from oracle import unet_ref, view_fusion_ref as vfr
    from view_fusion_amd import drivers
    vf = make_vf(TINY, SCHED_C1, dev, True)
    sd = {k: v.detach().cpu() for k, v in vf.denoise_fn.state_dict().items()}
    sched = vfr.schedule_buffers(vfr.beta_schedule(**SCHED_C1))
    fn = lambda x, a, l: unet_ref.unet_forward(sd, TINY, x, a, l)
    g = torch.Generator().manual_seed(400)
    # extrapolate: B=3, view counts in 7..23
    cond, angle = torch.rand(3, 23, 3, 16, 16, generator=g), torch.rand(3, 1, generator=g) * 6
    vc = torch.tensor([23, 7, 15])
    y_T, z_seq = torch.randn(3, 3, 16, 16, generator=g), torch.randn(10, 3, 3, 16, 16, generator=g)
    ret, logit_arr, weight_arr, _ = drivers.extrapolate(vf, cond.to(dev), angle.to(dev), view_count=vc,
                                                        y_t=y_T.to(dev), z_seq=z_seq.to(dev))
    with torch.no_grad():
        _, retr, lr, wr, _ = vfr.generate(fn, sched, cond, vc, angle, y_T, z_seq)
    np.testing.assert_allclose(ret.cpu().numpy(), retr.clamp(0, 1).numpy(), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(logit_arr.cpu().numpy(), lr.numpy(), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(weight_arr.cpu().numpy(), wr.numpy(), rtol=1e-4, atol=1e-5)
    # autoregressive rollout, 3 counts
    first = torch.rand(2, 3, 16, 16, generator=g)
    yts = [torch.randn(2, 3, 16, 16, generator=g) for _ in range(3)]
    zs = [torch.randn(10, 2, 3, 16, 16, generator=g) for _ in range(3)]
    orbit = drivers.autoregressive_rollout(vf, first.to(dev), steps=3, y_t=[t.to(dev) for t in yts],
                                           z_seq=[z.to(dev) for z in zs])
    c, want = first[:, None], []
    with torch.no_grad():
        for count in range(1, 4):
            ang = torch.full((2, 1), 2 * np.pi / 24 * count)
            *_, smp = vfr.generate(fn, sched, c, torch.full((2,), count), ang, yts[count - 1], zs[count - 1])
            c = torch.cat((c, smp[:, None]), dim=1)
            want.append(smp)
    np.testing.assert_allclose(orbit.cpu().numpy(), torch.stack(want, 1).numpy(), rtol=1e-4, atol=1e-4)
    # weight animation: one object, 24 target angles, 6 conditioning views each
    views = torch.rand(24, 3, 16, 16, generator=g)
    y_T, z_seq = torch.randn(24, 3, 16, 16, generator=g), torch.randn(10, 24, 3, 16, 16, generator=g)
    ret, logit_arr, weight_arr, cond_views, angles = drivers.orbit_frames(vf, views.to(dev), y_t=y_T.to(dev),
                                                                          z_seq=z_seq.to(dev))
    assert cond_views.shape == (24, 6, 3, 16, 16) and weight_arr.shape == (24, 10, 6, 3, 16, 16)
    want_cond = torch.stack([views[::4]] * 24)
    want_ang = torch.tensor([2 * np.pi / 24 * i for i in range(24)], dtype=torch.float32).unsqueeze(1)
    assert torch.equal(cond_views.cpu(), want_cond) and torch.equal(angles.cpu(), want_ang)
    with torch.no_grad():
        _, retr, lr, wr, _ = vfr.generate(fn, sched, want_cond, torch.full((24,), 6), want_ang, y_T, z_seq)
    np.testing.assert_allclose(ret.cpu().numpy(), retr.clamp(0, 1).numpy(), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(weight_arr.cpu().numpy(), wr.numpy(), rtol=1e-4, atol=1e-5)


def test_device_view_count_is_resolved_once(dev):
    """A device-tensor view_count (what the reference's loops hand over) costs one read-back per tensor, not one per
    p_sample call."""
    from view_fusion_amd import ops
    vc = torch.tensor([2, 1, 3], device=dev)
    a = ops.view_offsets(vc, dev)
    b = ops.view_offsets(vc, dev)
    assert a is b and a[1:] == (6, 3) and a[0].cpu().tolist() == [0, 2, 3, 6]
    vc += 1                                               # in-place change bumps the version: re-read
    c = ops.view_offsets(vc, dev)
    assert c[1:] == (9, 4)
    assert ops.view_offsets(torch.tensor([2, 1, 3]), dev)[1:] == (6, 3)
    with pytest.raises(ValueError):
        ops.view_offsets([2, 0], dev)


@pytest.mark.parametrize("hp,S", [
    (dict(in_channel=6, out_channel=6, inner_channel=64, norm_groups=32, channel_mults=(1, 1, 2, 2), attn_res=(16,),
          res_blocks=1, image_size=128), 2),          # the reference's default image size: 128x128 ... 16x16 maps
          # (>= 2 channels per group everywhere: with one, biases in front of a GroupNorm have analytically-zero
          # gradients that are pure round-off in the oracle and grow with the map size)
    (dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=16, channel_mults=(1, 2), attn_res=(32, 16),
          res_blocks=2, image_size=32), 5),           # out_channel 3 (no-weighting configs), 16 groups, attention at 32x32
])
def test_other_geometries_vs_oracle(dev, hp, S):
    """Constructor envelopes beyond the shipped YAMLs: image_size 128 (the reference's default, unet.py:20) and 32,
    another group count, attention on a 32x32 map (L = 1024: the generic attention path), out_channel 3 --
    forward + every gradient vs the oracle."""
    from oracle import unet_ref
    net = make_unet(hp, dev)
    g = torch.Generator().manual_seed(77)
    hw = hp["image_size"]
    x = torch.rand(S, 6, hw, hw, generator=g)
    angle = 2 * np.pi / 24 * torch.randint(0, 24, (S, 1), generator=g).float()
    level = torch.rand(S, 1, generator=g)
    xg = x.to(dev).requires_grad_(True)
    y = net(xg, angle.to(dev), level.to(dev))
    gy = torch.randn(y.shape, generator=g)
    (y * gy.to(dev)).sum().backward()
    sd = cpu_sd(net)
    xc = x.clone().requires_grad_(True)
    yc = unet_ref.unet_forward(sd, hp, xc, angle, level)
    (yc * gy).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), yc.detach().numpy(), rtol=1e-4, atol=5e-5)
    assert float((xg.grad.cpu() - xc.grad).norm() / xc.grad.norm()) < 1e-4
    for k, p in net.named_parameters():
        a, b = p.grad.detach().cpu().double(), sd[k].grad.double()
        assert float((a - b).norm()) <= 1e-4 * float(b.norm()) + 3e-5 * b.numel() ** 0.5, k
