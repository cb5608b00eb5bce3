#!/usr/bin/env python3
"""F(4x4,3x3) forward/dgrad kernel (winograd44f.hip) against the nested F(2,3)xF(4,3) kernel (winograd24.hip) on the
stride-1 3x3 layer shapes of the small UNet's 64x64 / 32x32 levels, S = 96 (GPU box): us per launch (HIP events,
best of 5 x 10 back-to-back launches through the C ABI) and the difference of the two results.
    python tools/wino44_table.py [S]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import _lib, ops  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device("cuda:0")
SHAPES = [(64, 64, 64, 0), (128, 64, 64, 0), (192, 64, 64, 0), (64, 128, 64, 0), (64, 192, 64, 0), (128, 128, 64, 2),
          (6, 64, 64, 0), (64, 6, 64, 0),
          (128, 128, 32, 0), (256, 128, 32, 0), (320, 128, 32, 0), (192, 128, 32, 0), (64, 128, 32, 0),
          (128, 256, 32, 0), (128, 320, 32, 0), (192, 192, 32, 2)]


def run(kind, x, w, Cin, Cout, H, m):
    abi = ops._WINO_ABI[kind]
    nf, nb = ctypes.c_long(), ctypes.c_long()
    _lib.call(abi[1], Cout, Cin, ctypes.byref(nf), ctypes.byref(nb))
    uf = torch.empty(nf.value, device=dev)
    ub = torch.empty(nb.value, device=dev)
    st = ops._stream()
    _lib.call(abi[2], ops._ptr(w), ops._ptr(uf), ops._ptr(ub), Cout, Cin, st)
    need = getattr(_lib.load(), abi[4])(S, Cin, Cout, H, H)
    ws = torch.empty(max(int(need), 1), device=dev)
    y = torch.empty(S, Cout, H, H, device=dev)
    args = (ops._ptr(x), ops._ptr(uf), None, None, None, ops._ptr(y), ops._ptr(ws) if need > 0 else None, int(need), S, Cin, Cout,
            H, H, m, st)
    for _ in range(3):
        _lib.call(abi[3], *args)
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            _lib.call(abi[3], *args)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    return y, best


NESTED_ONLY = os.environ.get("NESTED_ONLY") == "1"      # A/B of the nested kernel alone, incl. the 16x16 / 8x8 layers
if NESTED_ONLY:
    SHAPES = [(64, 64, 64, 0), (128, 128, 32, 0), (192, 192, 16, 0), (384, 192, 16, 0), (512, 192, 16, 0), (320, 192, 16, 0),
              (192, 320, 8, 0) if False else (128, 192, 16, 0), (320, 320, 16, 2), (320, 320, 8, 0), (640, 320, 8, 0), (512, 320, 8, 0)]
tot = [0.0, 0.0]
for Cin, Cout, H, m in SHAPES:
    g = torch.Generator().manual_seed(1)
    Hs = H // 2 if m == 2 else H
    x = torch.randn(S, Cin, Hs, Hs, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev)
    y1, t1 = run(1, x, w, Cin, Cout, H, m)
    y2, t2 = run(2, x, w, Cin, Cout, H, m) if not NESTED_ONLY else (y1, t1)
    # fp64 reference on two views
    xs = x[:2].double().cpu()
    if m == 2:
        xs = torch.nn.functional.interpolate(xs, scale_factor=2, mode="nearest")
    ref = torch.nn.functional.conv2d(xs, w.double().cpu(), padding=1)
    e1 = float((y1[:2].double().cpu() - ref).norm() / ref.norm())
    e2 = float((y2[:2].double().cpu() - ref).norm() / ref.norm())
    fl = 2.0 * S * Cout * Cin * 9 * H * H
    tot[0] += t1
    tot[1] += t2
    print(f"{Cin:4d}->{Cout:4d} @{H:2d} m={m}: nested {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TF direct-eq)   F(4x4) {t2:7.1f} us "
          f"({fl / t2 / 1e6:6.1f} TF)   ratio {t2 / t1:.3f}   rel-L2 vs fp64: nested {e1:.2e}  F(4x4) {e2:.2e}", flush=True)
print(f"sum: nested {tot[0]:.0f} us, F(4x4) {tot[1]:.0f} us, ratio {tot[1] / tot[0]:.3f}")
