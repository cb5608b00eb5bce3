"""Small host-side helpers shared by the harness, the tests and the fixture generator."""
import math

import torch


def deterministic_fill_(state_dict, base_seed=1000, skip=()):
    """Overwrite every floating tensor of `state_dict` IN PLACE with values that depend
    only on (position in the dict, shape).

    Used to give the reference (in the build container) and this package (anywhere)
    identical weights without shipping 136 MB of parameters: tensor #i is drawn from
    `torch.Generator().manual_seed(base_seed + i)` on the CPU as
      * ndim > 1  : U(-1,1) / sqrt(fan_in)          (fan_in = prod(shape[1:]))
      * 1-D *.weight (norm scales): 1 + 0.1 * U(-1,1)
      * 1-D otherwise (biases):     0.1 * U(-1,1)
    Keys listed in `skip` (e.g. schedule buffers) are left untouched.
    """
    for i, (k, v) in enumerate(state_dict.items()):
        if k in skip or not torch.is_floating_point(v):
            continue
        g = torch.Generator().manual_seed(base_seed + i)
        u = torch.rand(v.shape, generator=g, dtype=torch.float32) * 2 - 1
        if v.ndim > 1:
            u = u / math.sqrt(float(v[0].numel()))
        elif k.endswith("weight"):
            u = 1 + 0.1 * u
        else:
            u = 0.1 * u
        v.copy_(u.to(v.dtype))
    return state_dict


def tensor_digest(t, nsamples=64):
    """(sum, l2, absmax, strided samples) of a tensor, as float64 numpy-friendly values."""
    f = t.detach().double().reshape(-1).cpu()
    n = f.numel()
    step = max(1, n // nsamples)
    return dict(sum=float(f.sum()), l2=float(f.norm()), absmax=float(f.abs().max()),
                samples=f[::step][:nsamples].clone().numpy())
