"""CPU-only checks of the product's host side: schedules (exact vs the reference vectors),
LR schedule, ragged-offset logic, ViewFusion state_dict layout, the C-ABI surface, loud failure
without a GPU."""
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, TINY
from view_fusion_amd import UNet, ViewFusion, schedule, train

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


@pytest.mark.parametrize("name", sorted(SCHED))
def test_product_schedule_buffers_exact(name):
    g = np.load(os.path.join(GOLDEN, "schedules.npz"))
    vf = ViewFusion(None, {"train": SCHED[name]})
    vf.set_new_noise_schedule(device=torch.device("cpu"), phase="train")
    assert vf.num_timesteps == SCHED[name]["num_timesteps"]
    for k in schedule.BUFFER_NAMES:
        np.testing.assert_array_equal(getattr(vf, k).numpy(), g[f"{name}.{k}"], err_msg=k)


def test_unknown_schedule_raises_like_reference():
    with pytest.raises(NotImplementedError):
        schedule.make_beta_schedule("nope", 10)


def test_view_fusion_state_dict_matches_reference():
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["tiny"]
    vf = ViewFusion(UNet(**TINY), {"train": SCHED["linear_train"]})
    vf.set_new_noise_schedule(device=torch.device("cpu"), phase="train")
    mine = [[k, list(v.shape)] for k, v in vf.state_dict().items()]
    assert mine == ref
    # re-setting the schedule (train -> test phase) replaces the buffers, it does not duplicate them
    vf.beta_schedule["test"] = SCHED["linear_test"]
    vf.set_new_noise_schedule(device=torch.device("cpu"), phase="test")
    assert vf.gammas.shape == (1000,) and len(vf.state_dict()) == len(ref)


def test_lr_scheduler_matches_reference_formula():
    s = train.LrScheduler(peak_lr=1e-4, peak_it=2500, decay_it=4000000, decay_rate=0.16)
    assert s.get_cur_lr(0) == 0.0
    assert s.get_cur_lr(1250) == pytest.approx(5e-5)
    assert s.get_cur_lr(2500) == pytest.approx(1e-4)
    assert s.get_cur_lr(2500 + 4000000) == pytest.approx(1.6e-5)


def test_view_offsets_host_logic():
    from view_fusion_amd import ops
    off, S, mx = ops.view_offsets(torch.tensor([1, 3, 2]), torch.device("cpu"))
    assert off.dtype == torch.int32 and off.tolist() == [0, 1, 4, 6] and (S, mx) == (6, 3)
    off, S, mx = ops.view_offsets([2, 2], torch.device("cpu"))
    assert off.tolist() == [0, 2, 4] and (S, mx) == (4, 2)
    with pytest.raises(ValueError):
        ops.view_offsets([2, 0], torch.device("cpu"))


def test_c_abi_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from view_fusion_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "vf_hip.h")).read()
    declared = set(re.findall(r"\b(?:int|long)\s+(vf_\w+)\s*\(", header))
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_no_cpu_fallback():
    from view_fusion_amd._lib import VFHipError
    vf = ViewFusion(UNet(**TINY), {"train": SCHED["linear_train"]})
    vf.set_new_noise_schedule(device=torch.device("cpu"), phase="train")
    with pytest.raises(VFHipError):
        vf(torch.rand(2, 2, 3, 16, 16), torch.tensor([2, 2]), torch.rand(2, 1), y_0=torch.rand(2, 3, 16, 16))
    with pytest.raises(VFHipError):
        UNet(**TINY)(torch.rand(1, 6, 16, 16), torch.rand(1, 1), torch.rand(1, 1))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "view_fusion_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "import oracle" not in src and "from oracle" not in src, f


def test_synthetic_batch_shapes_and_per_rank_seeds():
    a = train.synthetic_batch(4, 6, 16, "cpu", seed=0)
    b = train.synthetic_batch(4, 6, 16, "cpu", seed=1)
    assert a["y_0"].shape == (4, 3, 16, 16) and a["y_cond"].shape == (4, 6, 3, 16, 16)
    assert a["angle"].shape == (4, 1) and a["view_count"].tolist() == [6] * 4
    assert not torch.equal(a["y_0"], b["y_0"])
    r = train.synthetic_batch(64, 6, 8, "cpu", seed=0, ragged=True)["view_count"]
    assert int(r.min()) >= 1 and int(r.max()) <= 6 and len(set(r.tolist())) > 1
    assert float(a["y_0"].min()) >= 0 and float(a["y_0"].max()) <= 1


def test_checkpoint_wire_format_roundtrip(tmp_path):
    """utils/checkpoint.py layout: {"model", "optimizer", it, t, run_id, ssim, psnr}."""
    from view_fusion_amd import drivers
    vf = ViewFusion(UNet(**TINY), {"train": SCHED["linear_train"]})
    vf.set_new_noise_schedule(device=torch.device("cpu"), phase="train")
    opt = torch.optim.Adam(vf.parameters(), lr=1e-4)
    path = str(tmp_path / "logs" / "model.pt")
    drivers.save_checkpoint(path, vf, opt, it=7, t=1.5, run_id="abc", ssim=0.25, psnr=11.0)
    raw = torch.load(path, weights_only=False)
    assert set(raw) == {"model", "optimizer", "it", "t", "run_id", "ssim", "psnr"}
    assert list(raw["model"]) == list(vf.state_dict())
    vf2 = ViewFusion(UNet(**TINY), {"train": SCHED["linear_train"]})
    vf2.set_new_noise_schedule(device=torch.device("cpu"), phase="train")
    rest = drivers.load_checkpoint(path, vf2, torch.optim.Adam(vf2.parameters(), lr=1e-4))
    assert rest == dict(it=7, t=1.5, run_id="abc", ssim=0.25, psnr=11.0)
    assert all(torch.equal(a, b) for a, b in zip(vf.state_dict().values(), vf2.state_dict().values()))


def test_reduce_dict_identity_without_process_group():
    from view_fusion_amd import drivers
    d = {"psnr": torch.tensor(3.0)}
    assert drivers.reduce_dict(d) is d


def test_drivers_run_in_eval_mode_and_restore_the_mode():
    """Experiment.eval calls model.eval() first (experiment.py:316): the sampler drivers switch Dropout off for the call
    and put the model back into the mode they found it in."""
    import torch
    from view_fusion_amd import drivers

    class Probe(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.drop = torch.nn.Dropout(0.5)
            self.seen = []

        def forward(self, **kw):
            self.seen.append(self.training)
            b = kw["y_cond"].shape[0]
            z = torch.zeros(b, 2, 3, 4, 4)
            return z[:, 0], z, None, None, z[:, 0]

    m = Probe().train()
    cond = torch.zeros(2, 23, 3, 4, 4)
    drivers.extrapolate(m, cond, torch.zeros(2, 1), view_count=torch.tensor([7, 8]))
    assert m.seen == [False] and m.training
    m.eval()
    drivers.extrapolate(m, cond, torch.zeros(2, 1), view_count=torch.tensor([7, 8]))
    assert m.seen == [False, False] and not m.training


def test_trainer_restores_training_mode_of_submodules():
    """Trainer.step re-enters training mode when only a SUBMODULE with a mode-dependent layer was put into eval mode
    (the reference calls model.train() every iteration, experiment.py:286)."""
    import torch
    from view_fusion_amd import train
    from view_fusion_amd.unet import _ResBlock

    blk = _ResBlock(4, 4, 8, 2, dropout=0.1)
    root = torch.nn.Sequential(blk)
    tr = train.Trainer.__new__(train.Trainer)
    tr._mode_modules = [m for m in root.modules() if getattr(m, "dropout", 0) and hasattr(m, "_drop")]
    assert tr._mode_modules == [blk]
    blk.eval()
    assert root.training and any(not m.training for m in tr._mode_modules)


def test_trainer_step_graph_gating_is_host_logic():
    """Trainer(graph=True): a CPU model never takes the graph path; the geometry key is the tensor shapes plus the SUM
    of view_count (the per-sample offsets are a graph input), and everything that cannot be replayed maps to None."""
    import torch
    from view_fusion_amd import train
    from test_ddp_gloo import StandIn

    tr = train.Trainer(StandIn(), graph=True, lr_warmup=1)
    assert tr.use_graph is False                       # CPU parameters: eager by construction
    bt = train.synthetic_batch(4, 3, 8, "cpu", seed=0)
    for _ in range(4):
        loss = tr.step(bt)
    assert tr.graph_steps == 0 and not loss.requires_grad      # the eager step returns a detached loss

    tr.use_graph = True                                # the key function itself (no GPU work involved)

    class Fake(torch.Tensor):                          # a CPU tensor that claims to live on the GPU
        is_cuda = property(lambda self: True)

    f = lambda t: t.as_subclass(Fake)
    gb = dict(y_0=f(bt["y_0"]), y_cond=f(bt["y_cond"]), angle=f(bt["angle"]))
    k1, vc1 = tr._graph_key(dict(gb, view_count=torch.tensor([1, 2, 3, 3])), {})
    k2, vc2 = tr._graph_key(dict(gb, view_count=[3, 3, 2, 1]), {})
    assert k1 == k2 and k1[-1] == 9 and vc1 == (1, 2, 3, 3) and vc2 == (3, 3, 2, 1)
    k3, _ = tr._graph_key(dict(gb, view_count=[3, 3, 3, 1]), {})
    assert k3 != k1
    assert tr._graph_key(dict(gb, view_count=[0, 3, 3, 3]), {}) is None          # the eager path raises the error
    assert tr._graph_key(dict(gb, view_count=[4, 3, 1, 1]), {}) is None          # more views than y_cond holds
    assert tr._graph_key(dict(gb, view_count=f(torch.tensor([1, 2, 3, 3]))), {}) is None    # device-resident counts
    assert tr._graph_key(dict(gb, view_count=[1, 2, 3, 3]), {"generate": True}) is None
    assert tr._graph_key(dict(gb, view_count=[1, 2, 3, 3]), {"t": torch.zeros(4)}) is None   # a CPU draw
    kt, _ = tr._graph_key(dict(gb, view_count=[1, 2, 3, 3]), {"t": f(torch.zeros(4))})
    assert kt != k1 and kt[-1] == 9
    assert tr._graph_key(dict(bt, view_count=[1, 2, 3, 3]), {}) is None          # CPU images


def test_multi_rank_agreement_points_are_host_logic():
    """train.Trainer._agree: the failure flag of the gradient arena is read at iterations every rank
    computes alike -- 1, 4, 16 after the last change of mode, then every AGREE_EVERY-th -- so a capture that fails late
    (a new ragged geometry at iteration 1000, a run resumed at it = 100000) is seen within AGREE_EVERY iterations; a
    raised flag steps the mode down once and restarts the schedule."""
    class Flag:
        def __init__(self):
            self.v, self.reads = 0.0, []

        def item(self):
            self.reads.append(tr.it)
            return self.v

        def zero_(self):
            self.v = 0.0

    class Arena:
        capturable, flag_value, flat = False, 0.0, None

    tr = train.Trainer(torch.nn.Linear(2, 2), world=1, lr_warmup=1, graph=False)
    tr.arena, tr._graph_wanted, tr.use_graph = Arena(), True, True
    tr.arena.flag_acc = f = Flag()
    base = tr._check_base
    for tr.it in range(base, base + 300):
        tr._agree()
    assert f.reads[:3] == [base + 1, base + 4, base + 16]
    assert f.reads[3:] == [base + k for k in range(64, 300, 64)]
    # resumed far into a run: the next read is at most AGREE_EVERY iterations away
    f.reads.clear()
    for tr.it in range(100000, 100000 + 2 * train.Trainer.AGREE_EVERY):
        tr._agree()
    assert f.reads and f.reads[0] - 100000 < train.Trainer.AGREE_EVERY and len(f.reads) == 2
    # a raised flag: one demotion (split -> eager), flag cleared, schedule restarts at that iteration
    f.v = 0.5
    tr.it = f.reads[-1] + train.Trainer.AGREE_EVERY
    tr._agree()
    assert tr.demotions == 1 and tr.mode == "eager" and f.v == 0.0 and tr._check_base == tr.it
    assert train.Trainer.inject_capture_failure is None and "VF_TEST_FAIL_CAPTURE" not in open(train.__file__).read()


def test_winograd_kernel_choice_is_host_logic():
    """ops.wino_kind (round 4): which of the three conv paths a stride-1 3x3 layer takes is decided on the host from the
    library's tile plans and a cost model -- no GPU involved.  At the bench geometry (S = 96) the 64x64 layers and the
    deep 32x32 layers take the F(4x4,3x3) kernel, 128 -> 128 at 32x32 (1.5 rounds of tiles with a short K) and the
    16x16 / 8x8 maps stay on the nested kernel, one or two views (the sampler) on the direct kernel; the decision is the
    same for the pack (made outside autograd.Function) and the launch (made inside, where grad mode is off) because
    both pass `train` explicitly; forcing flags override it."""
    from view_fusion_amd import ops
    k = lambda S, ci, co, h, train=True, m=0: ops.wino_kind(S, ci, co, h, h, 3, m, train)
    assert [k(96, 64, 64, 64), k(96, 128, 64, 64), k(96, 192, 64, 64), k(96, 6, 64, 64), k(96, 64, 6, 64)] == [2] * 5
    assert [k(96, 256, 128, 32), k(96, 320, 128, 32), k(96, 128, 128, 64, m=2), k(96, 192, 192, 32, m=2)] == [2] * 4
    assert k(96, 128, 128, 32) == 1 and k(96, 192, 192, 16) == 1 and k(96, 320, 320, 8) == 1 and k(96, 640, 320, 8) == 1
    assert k(1, 64, 64, 64) == 0 and k(2, 320, 320, 8) == 0
    assert ops.wino_kind(96, 64, 64, 64, 64, 1, 0, True) == 0 and ops.wino_kind(96, 64, 64, 64, 64, 3, 1, True) == 0
    assert k(96, 64, 64, 64, train=False) == 2               # forward alone (the sampler at B = 16)
    import torch
    with torch.no_grad():                                   # default `train` = whether autograd records
        assert ops.wino_kind(96, 64, 64, 64, 64, 3, 0) == ops.wino_kind(96, 64, 64, 64, 64, 3, 0, False)
    ops.st.FORCE_WINOGRAD = True
    try:
        assert k(2, 64, 64, 64) == 1 and k(96, 64, 64, 64) == 1
        ops.st.FORCE_WINOGRAD44 = True
        assert k(2, 64, 64, 64) == 2 and k(2, 192, 192, 16) == 1          # (no F(4x4) kernel for 16x16 maps)
    finally:
        ops.st.FORCE_WINOGRAD = ops.st.FORCE_WINOGRAD44 = False
    assert ops.use_winograd(96, 64, 64, 64, 64, 3, 0, True) and not ops.use_winograd(1, 64, 64, 64, 64, 3, 0, False)


def test_deferred_slab_sum_table_is_host_logic():
    """Round 5: the rows of the one slab-sum launch per backward pass (ops._wred_rows).  Each row is what the main kernel's
    launcher wrote (9 x int64: five pointers, then int32 pairs); the host only fills the row's `first` field -- the low half
    of word 8 -- with the running sum of the preceding rows' workgroup counts, and leaves everything else alone."""
    from view_fusion_amd import ops
    mk = lambda tag, pad: ([tag + i for i in range(8)] + [pad << 32], )
    pend = [(mk(100, 0)[0], 65, None), (mk(200, 7)[0], 1280, None), (mk(300, 0)[0], 3, None)]
    rows, total = ops._wred_rows(pend)
    assert total == 65 + 1280 + 3
    assert [r[8] & 0xFFFFFFFF for r in rows] == [0, 65, 65 + 1280]
    assert [r[8] >> 32 for r in rows] == [0, 7, 0]                    # the upper half (padding) is not touched
    assert [r[:8] for r in rows] == [p[0][:8] for p in pend]
    assert pend[1][0][8] == 7 << 32                                  # the registered rows themselves stay as they were


def test_attention_kernel_choice_mirrors_the_launcher():
    """The L = 256 attention forward picks between 8 S workgroups of the 32-query kernel (three per compute unit) and 2 S
    workgroups of the 128-query kernel by a two-term cost model (attention.hip, vf_attention_fwd); this restates it so that a
    change of the constants is a conscious one: the training batch S = 96 and the C4 batch S = 48 must take the 32-query
    kernel, S = 128 (2 S = 256 workgroups: one per compute unit) the 128-query one."""
    q32_wins = lambda S: 177 * ((S + 31) // 32) + 30 < 660 * ((S + 127) // 128)
    assert all(q32_wins(S) for S in (17, 32, 48, 64, 96, 144, 160, 192, 224))
    assert not any(q32_wins(S) for S in (97, 100, 112, 128, 256))
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "view_fusion_amd", "csrc",
                            "attention.hip")).read()
    assert "177 * ((S + 31) / 32) + 30 < 660 * ((S + 127) / 128)" in src
