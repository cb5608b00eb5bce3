#!/usr/bin/env python3
"""Time the attention BACKWARD (four batched products + the softmax backward, ops.attention) at training batch sizes and print
the launches it made.   python tools/one_attn_bwd.py [S ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (H, C) in ((16, 192), (8, 320)):
    L = H * H
    for S in ([int(a) for a in sys.argv[1:]] or (48, 96)):
        qkv = (torch.randn(S, 3 * C, H, H) * 2).to(dev).requires_grad_(True)
        gy = torch.randn(S, C, H, H).to(dev)
        out = ops.attention(qkv)

        def bwd():
            qkv.grad = None
            out.backward(gy, retain_graph=True)
        for _ in range(3):
            bwd()
        ops.st.KERNEL_LOG = []
        bwd()
        torch.cuda.synchronize()
        names = [e[5] for e in ops.st.KERNEL_LOG]
        ops.st.KERNEL_LOG = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for _ in range(n):
            bwd()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        fl = 4 * 2.0 * S * L * L * C
        print(f"L={L} C={C} S={S:3d}  backward {us:7.1f} us  {fl / us / 1e6:6.1f} TF = {fl / us / 1e6 / 157.3:.2f} of the fp32 MFMA peak   "
              f"launches: {names}")
