#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase clocks of conv_mfma_kernel (prologue / K loop / epilogue).
Builds a private libvf_hip_stamps.so with -DVF_CONV_STAMPS; never used by the product.
usage: conv_stamps.py Cin Cout H KS [S]"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag_build import diag_build  # noqa: E402
so = diag_build("libvf_hip_stamps.so", ["conv.hip", "norm.hip"], ["-DVF_CONV_STAMPS"])   # built here (no GPU needed) or on the box
if len(sys.argv) < 5:
    sys.exit(0)
Cin, Cout, H, KS = (int(v) for v in sys.argv[1:5])
S = int(sys.argv[5]) if len(sys.argv) > 5 else 96
lib = ctypes.CDLL(so)
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
lib.vf_conv_pack_sizes.argtypes = [I, I, I, ctypes.POINTER(L), ctypes.POINTER(L)]
lib.vf_conv_pack_weights.argtypes = [P, P, P, I, I, I, P]
lib.vf_conv_fwd.argtypes = [P, P, P, P, P, P, P, L, I, I, I, I, I, I, I, P]
dev = torch.device("cuda:0")
w = torch.randn(Cout, Cin, KS, KS, device=dev) / (Cin * KS * KS) ** 0.5
x = torch.rand(S, Cin, H, H, device=dev)
y = torch.empty(S, Cout, H, H, device=dev)
nf, nb = L(), L()
lib.vf_conv_pack_sizes(Cout, Cin, KS, ctypes.byref(nf), ctypes.byref(nb))
wf = torch.empty(nf.value, device=dev); wb = torch.empty(nb.value, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.vf_conv_pack_weights(w.data_ptr(), wf.data_ptr(), wb.data_ptr(), Cout, Cin, KS, st)
stamps = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
for _ in range(3):   # ws only carries the stamps here: nblk >= 384 so split-K stays off
    rc = lib.vf_conv_fwd(x.data_ptr(), wf.data_ptr(), None, None, None, y.data_ptr(), stamps.data_ptr(), 0, S, Cin,
                         Cout, H, H, KS, 0, st)
    assert rc == 0
torch.cuda.synchronize()
a = stamps.cpu().numpy().reshape(-1, 8)
a = a[a[:, 3] != 0]
n = len(a)
pro, loop, epi = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
wall = (a[:, 5].max() - a[:, 4].min()) / 100e6          # s_memrealtime = 100 MHz
life = (a[:, 5] - a[:, 4]) / 100e6
clk = (a[:, 3] - a[:, 0]).astype(np.float64) / np.maximum((a[:, 5] - a[:, 4]) / 100e6, 1e-9) / 1e9
flops = 2.0 * S * Cout * Cin * KS * KS * H * H
print(f"shape Cin={Cin} Cout={Cout} H={H} KS={KS} S={S}: {n} workgroups, kernel wall {wall * 1e6:.1f} us -> {flops / wall / 1e12:.1f} TF")
print(f"  cycles/WG  prologue {pro.mean():9.0f}  K-loop {loop.mean():9.0f}  epilogue {epi.mean():9.0f}   (share of lifetime: "
      f"{pro.sum() / (pro + loop + epi).sum():.3f} / {loop.sum() / (pro + loop + epi).sum():.3f} / {epi.sum() / (pro + loop + epi).sum():.3f})")
print(f"  WG lifetime mean {life.mean() * 1e6:.1f} us, shader clock while resident {np.median(clk):.2f} GHz, avg concurrent WGs/CU {life.sum() / wall / 256:.2f}")
t0 = (a[:, 4] - a[:, 4].min()) / 100e6
t1 = (a[:, 5] - a[:, 4].min()) / 100e6
grid = np.linspace(0, wall, 11)[:-1]
live = [int(((t0 <= g + wall / 20) & (t1 > g + wall / 20)).sum()) for g in grid]
print(f"  started within first 5% of wall: {int((t0 < 0.05 * wall).sum())} / {n};  live WGs at 5%,15%..95% of wall: {live}")
cuid = (a[:, 6] >> 8) & 0xFF          # HW_ID[15:8] = se_id | sh_id | cu_id  (per XCD)
print(f"  distinct (se,sh,cu) ids: {len(np.unique(cuid))}")
