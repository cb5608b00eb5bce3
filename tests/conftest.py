import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

TINY = dict(in_channel=6, out_channel=6, inner_channel=32, norm_groups=32,
            channel_mults=(1, 2), attn_res=(8,), res_blocks=1, image_size=16)
SMALL = dict(in_channel=6, out_channel=6, inner_channel=64, norm_groups=32,
             channel_mults=(1, 2, 3, 5), attn_res=(16,), res_blocks=3, image_size=64)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

MICRO = dict(in_channel=6, out_channel=6, inner_channel=8, norm_groups=8,
             channel_mults=(1, 2), attn_res=(8,), res_blocks=1, image_size=16)
SCHED_TRAIN = dict(schedule="linear", num_timesteps=2000, linear_start=1e-6, linear_end=1e-2)
SCHED_TEST = dict(schedule="linear", num_timesteps=1000, linear_start=1e-4, linear_end=0.09)
SCHED_C1 = dict(schedule="linear", num_timesteps=10, linear_start=1e-4, linear_end=0.09)


def golden_inputs(B, N, hw, seed):
    """The seeded input recipe of tests/golden/make_golden.py:inputs (fixtures that store seeds instead of tensors)."""
    import numpy as np
    import torch
    g = torch.Generator().manual_seed(seed)
    y_0 = torch.rand(B, 3, hw, hw, generator=g)
    y_cond = torch.rand(B, N, 3, hw, hw, generator=g)
    angle = 2 * np.pi / 24 * torch.randint(0, 24, (B, 1), generator=g).float()
    noise = torch.randn(B, 3, hw, hw, generator=g)
    return y_0, y_cond, angle, noise


def c1_chain_inputs(g):
    """y_cond, angle, view_count, y_T, z_seq of the BASELINE-C1 fixture (c1_small_chain.npz stores only the seeds)."""
    import torch
    y_0, y_cond, angle, noise = golden_inputs(2, 2, 64, int(g["seed_inputs"]))
    y_T = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(int(g["seed_yT"])))
    gz = torch.Generator().manual_seed(int(g["seed_z"]))
    # make_golden draws z from the GLOBAL generator after manual_seed: same stream as a fresh generator of that seed
    z = [torch.randn(2, 3, 64, 64, generator=gz) for _ in range(9)]
    z_seq = torch.stack([torch.zeros(2, 3, 64, 64)] + z[::-1])
    return y_0, y_cond, angle, noise, torch.tensor(g["view_count"]), y_T, z_seq


def check_digest(t, g, name, rtol=1e-4, atol=5e-5):
    """Compare a tensor with its (stat, samples, shape) digest in fixture g."""
    import numpy as np
    from view_fusion_amd.utils import tensor_digest
    assert tuple(t.shape) == tuple(g[f"{name}.shape"]), name
    d = tensor_digest(t, nsamples=256)
    ref = g[f"{name}.stat"]
    assert abs(d["l2"] - ref[1]) <= rtol * ref[1], (name, d["l2"], ref[1])
    np.testing.assert_allclose(d["samples"], g[f"{name}.samples"], rtol=rtol, atol=atol, err_msg=name)
