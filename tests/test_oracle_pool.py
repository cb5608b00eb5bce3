"""The sample-parallel oracle front ends (tests/oracle_pool.py) return what the serial oracle returns."""
import numpy as np
import torch

import oracle_pool
from conftest import TINY
from oracle import unet_ref, view_fusion_ref as vfr

SCHED = dict(schedule="linear", num_timesteps=12, linear_start=1e-4, linear_end=0.09)


def _setup(B=3, N=3, hw=16):
    from view_fusion_amd import UNet
    from view_fusion_amd.utils import deterministic_fill_
    net = UNet(**TINY)
    deterministic_fill_(net.state_dict())
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    y_cond = torch.rand(B, N, 3, hw, hw, generator=g)
    angle = 2 * np.pi / 24 * torch.randint(0, 24, (B, 1), generator=g).float()
    y_T = torch.randn(B, 3, hw, hw, generator=g)
    z_seq = torch.randn(12, B, 3, hw, hw, generator=g)
    vc = torch.tensor([3, 1, 2])
    return sd, y_cond, angle, y_T, z_seq, vc, g


def test_pool_generate_chain_and_train_match_serial_oracle():
    sd, y_cond, angle, y_T, z_seq, vc, g = _setup()
    fn = lambda x, a, l: unet_ref.unet_forward(sd, TINY, x, a, l)
    sched = vfr.schedule_buffers(vfr.beta_schedule(**SCHED))
    try:
        got = oracle_pool.generate(sd, TINY, SCHED, y_cond, vc, angle, y_T, z_seq, sample_num=4)
        with torch.no_grad():
            want = vfr.generate(fn, sched, y_cond, vc, angle, y_T, z_seq, 4)
        for a, b in zip(got, want):
            assert a.shape == b.shape
            np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=0, atol=2e-6)

        y, kept, w = oracle_pool.chain(sd, TINY, SCHED, y_cond, vc, angle, y_T, z_seq[:5], t_hi=7, t_lo=3, keep_every=2)
        yy, keep = y_T, []
        with torch.no_grad():
            for n, i in enumerate(range(7, 2, -1)):
                yy, _, ww = vfr.p_sample(fn, sched, yy, y_cond, vc, angle, torch.full((3,), i), z_seq[n])
                if (n + 1) % 2 == 0:
                    keep.append(yy)
        np.testing.assert_allclose(y.numpy(), yy.numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(kept.numpy(), torch.stack(keep).numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(w.numpy(), ww.numpy(), rtol=0, atol=2e-6)

        y_0, noise = torch.rand(3, 3, 16, 16, generator=g), torch.randn(3, 3, 16, 16, generator=g)
        t, u = torch.tensor([5, 1, 11]), torch.rand(3, 1, generator=g)
        loss, grads = oracle_pool.train(sd, TINY, SCHED, y_cond, vc, angle, y_0, t, u, noise, chunk=1)
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ref = vfr.train_loss(lambda x, a, l: unet_ref.unet_forward(sdg, TINY, x, a, l), sched, y_cond, vc, angle, y_0, t,
                             u, noise)
        ref.backward()
        assert abs(loss - ref.item()) <= 1e-6 * abs(ref.item())
        for k, v in sdg.items():
            if v.grad is not None:
                assert float((grads[k].float() - v.grad).norm()) <= 1e-5 * float(v.grad.norm()) + 1e-7, k
    finally:
        oracle_pool.close()
