#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase clocks of attn_fwd_kh_kernel / attn_fwd_q32_kernel (S views, L=256, C=192).
usage: attn_stamps.py [S] [q32]"""
import ctypes, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag_build import diag_build  # noqa: E402
so = diag_build("libvf_attn_stamps.so", ["attention.hip"], ["-DVF_ATTN_STAMPS"])   # rebuilt whenever attention.hip / a header changed
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 96
Q32 = len(sys.argv) > 2 and sys.argv[2] == "q32"        # the 32-query kernel (8 S workgroups) instead of the 128-query one
os.environ["VF_ATTN_Q32"] = "1" if Q32 else "0"
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
a = stamps.cpu().numpy()[: (8 if Q32 else 2) * S * 8].reshape(-1, 8)
life = (a[:, 7] - a[:, 6]) / 100e6
tot = (a[:, 4] - a[:, 0]).astype(float)
wall = (a[:, 7].max() - a[:, 6].min()) / 100e6
print(f"S={S}: {len(a)} WGs, wall {wall * 1e6:.1f} us, WG life {life.mean() * 1e6:.1f} us (min {life.min() * 1e6:.1f} max {life.max() * 1e6:.1f}), "
      f"shader clock {np.median(tot / life) / 1e9:.2f} GHz")
ph = [(a[:, i + 1] - a[:, i]).mean() for i in range(4)]
print(f"  cycles/WG: total {tot.mean():.0f} = QK {ph[0]:.0f} + softmax {ph[1]:.0f} + (P write skipped) {ph[2]:.0f} + PV {ph[3]:.0f};  "
      f"MFMA issue floor per SIMD and workgroup: QK {(12 * 16 if Q32 else 2 * 12 * 8 * 4) * 64} + PV {(12 * 16 if Q32 else 2 * 6 * 16 * 4) * 64}")
