#!/usr/bin/env python3
"""Diagnostic: per-wave phase clocks of attn_fwd_split_kernel<256> (few views: the sampler).  usage: attn_split_stamps.py [S]"""
import ctypes, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag_build import diag_build  # noqa: E402
so = diag_build("libvf_attn_stamps.so", ["attention.hip"], ["-DVF_ATTN_STAMPS"])
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L, C = 256, 192
lib = ctypes.CDLL(so)
P, I = ctypes.c_void_p, ctypes.c_int
lib.vf_attention_fwd.argtypes = [P, P, P, I, I, I, P]
dev = torch.device("cuda:0")
qkv = torch.randn(S, 3 * C, L, device=dev)
out = torch.empty(S, C, L, device=dev)
stamps = torch.zeros(S * L * L // 2, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    assert lib.vf_attention_fwd(qkv.data_ptr(), out.data_ptr(), stamps.data_ptr(), S, C, L, st) == 0
torch.cuda.synchronize()
a = stamps.cpu().numpy()[: S * 8 * 8 * 8].reshape(S * 8, 8, 8)          # [workgroup][wave][slot]
wall = (a[..., 7].max() - a[..., 6].min()) / 100e6
life = (a[..., 7] - a[..., 6]) / 100e6
tot = (a[..., 4] - a[..., 0]).astype(float)
print(f"S={S}: {a.shape[0]} workgroups x 8 waves, launch wall {wall * 1e6:.1f} us, wave life {life.mean() * 1e6:.1f} us "
      f"(max {life.max() * 1e6:.1f}), shader clock {np.median(tot / life) / 1e9:.2f} GHz")
ph = [(a[..., i + 1] - a[..., i]).astype(float) for i in range(4)]
names = ["QK (6 chunks: stage + 16 MFMA)", "softmax", "P -> LDS + barrier", "PV (waves 0-5: 128 MFMA)"]
for n, p in zip(names, ph):
    print(f"  {n:34s} mean {p.mean():8.0f} cycles   per wave of WG 0: " + " ".join(f"{v:6.0f}" for v in p[0]))
print(f"  MFMA issue: QK {96 * 64} cycles, PV {128 * 64} cycles per wave")
start = (a[..., 6] - a[..., 6].min()) / 100e6 * 1e6
print("  wave start offsets (us) of WG 0..7, wave 0: " + " ".join(f"{v:5.2f}" for v in start[:8, 0]))
