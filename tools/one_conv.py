#!/usr/bin/env python3
"""Run ONE conv layer shape (fwd + dgrad + wgrad) repeatedly -- target for rocprofv3 --pmc.
usage: one_conv.py Cin Cout H KS mode [S] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import ops  # noqa: E402

Cin, Cout, H, KS = (int(v) for v in sys.argv[1:5])
mode = sys.argv[5] if len(sys.argv) > 5 else "same"
S = int(sys.argv[6]) if len(sys.argv) > 6 else 96
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
dev = torch.device("cuda:0")
layer = torch.nn.Conv2d(Cin, Cout, KS, padding=KS // 2).to(dev)
x = torch.rand(S, Cin, H, H, device=dev, requires_grad=True)
Ho = H // 2 if mode == "down2" else (H * 2 if mode == "up2" else H)
gy = torch.rand(S, Cout, Ho, Ho, device=dev)
res = torch.rand(S, Cout, Ho, Ho, device=dev)
fl = 2.0 * S * Cout * Cin * KS * KS * Ho * Ho
ops.st.KERNEL_LOG = []
for it in range(reps + 2):
    if it == 2:
        ops.st.KERNEL_LOG.clear()
    y = ops.conv2d(x, layer, residual=res, mode=mode)
    y.backward(gy)
torch.cuda.synchronize()
agg = {}
for kind, flops, e0, e1, tag, _name in ops.st.KERNEL_LOG:
    a = agg.setdefault(kind, [0.0, 0]); a[0] += e0.elapsed_time(e1) * 1e-3; a[1] += 1
print(" ".join(f"{k}: {v[0] / v[1] * 1e6:.1f}us {fl / (v[0] / v[1]) / 1e12:.1f}TF" for k, v in agg.items()),
      f"| Cin={Cin} Cout={Cout} H={H} KS={KS} {mode} ablate={os.environ.get('VF_CONV_ABLATE', '0')}")
