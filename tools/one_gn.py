#!/usr/bin/env python3
"""GroupNorm+Swish forward/backward on one shape, repeated -- target for rocprofv3 --pmc and
HBM-roofline timing.  usage: one_gn.py C H [S] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import ops  # noqa: E402

C, H = int(sys.argv[1]), int(sys.argv[2])
S = int(sys.argv[3]) if len(sys.argv) > 3 else 96
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
x = torch.randn(S, C, H, H, device=dev, requires_grad=True)
g = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
gy = torch.randn(S, C, H, H, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for it in range(reps + 2):
    ev[0].record()
    y = ops.group_norm(x, g, b, 32, True)
    ev[1].record()
    y.backward(gy)
    ev[2].record()
    torch.cuda.synchronize()
    if it >= 2:
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
nbytes = x.numel() * 4
print(f"GN+Swish C={C} H={H} S={S}: fwd {tf / reps * 1e3:.1f} us = {2 * nbytes / (tf / reps * 1e-3) / 1e9:.0f} GB/s (algorithmic 8 B/elem) | "
      f"bwd(+colsum) {tb / reps * 1e3:.1f} us = {3 * nbytes / (tb / reps * 1e-3) / 1e9:.0f} GB/s (algorithmic 12 B/elem)")
