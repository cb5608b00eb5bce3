#!/usr/bin/env python3
"""Time vf_attention_fwd for the UNet's two attention shapes at sampler and training batch sizes, and check it
against a torch fp32 formula.   python tools/one_attn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
for (L, C) in ((256, 192), (64, 320)):
    for S in ((1, 6, 12, 17, 20, 24, 28, 32, 44, 48, 53, 56, 64, 72, 80, 88, 96, 100, 104, 112, 120, 128, 144, 160, 192, 256) if L == 256 else (1, 6, 12, 48, 96)):
        for wantP in (False, True):
            g = torch.Generator(device="cpu").manual_seed(S + L)
            qkv = torch.randn(S, 3 * C, L, generator=g).to(dev)
            out = torch.empty(S, C, L, device=dev)
            P = torch.empty(S, L, L, device=dev) if wantP else None
            args = (qkv.data_ptr(), out.data_ptr(), P.data_ptr() if wantP else None, S, C, L, None)
            for _ in range(3):
                _lib.call("vf_attention_fwd", *args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record()
            for _ in range(n):
                _lib.call("vf_attention_fwd", *args)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            q, k, v = qkv[:, :C].double(), qkv[:, C:2 * C].double(), qkv[:, 2 * C:].double()
            p = torch.softmax(torch.einsum("sci,scj->sij", q, k) / C ** 0.5, dim=-1)
            ref = torch.einsum("sij,scj->sci", p, v)
            err = float((out.double() - ref).abs().max())
            perr = float((P.double() - p).abs().max()) if wantP else 0.0
            print(f"L={L} C={C} S={S:3d} P={int(wantP)}  {us:7.1f} us  {4.0 * S * L * L * C / us / 1e6:6.1f} TF  "
                  f"max|dO|={err:.2e} max|dP|={perr:.2e}")
