#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase clocks of wino_conv_kernel.  usage: wino_stamps.py Cin Cout H [S]"""
import ctypes, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
abl = os.environ.get("ABL", "0")
so = os.path.join(ROOT, "build", f"libvf_wino_stamps{abl}.so")
if not os.path.exists(so):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-DVF_CONV_STAMPS", f"-DVF_WINO_ABL={abl}", "-I",
                    os.path.join(ROOT, "include"), os.path.join(ROOT, "view_fusion_amd/csrc/winograd.hip"), "-o", so], check=True)
if len(sys.argv) < 4:
    sys.exit(0)
Cin, Cout, H = (int(v) for v in sys.argv[1:4]); S = int(sys.argv[4]) if len(sys.argv) > 4 else 96
lib = ctypes.CDLL(so)
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
lib.vf_wino_pack_sizes.argtypes = [I, I, ctypes.POINTER(L), ctypes.POINTER(L)]
lib.vf_wino_pack_weights.argtypes = [P, P, P, I, I, P]
lib.vf_wino_conv_fwd.argtypes = [P, P, P, P, P, P, P, ctypes.c_long, I, I, I, I, I, I, P]
dev = torch.device("cuda:0")
w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
x = torch.rand(S, Cin, H, H, device=dev); y = torch.empty(S, Cout, H, H, device=dev)
nf, nb = L(), L(); lib.vf_wino_pack_sizes(Cout, Cin, ctypes.byref(nf), ctypes.byref(nb))
uf = torch.empty(nf.value, device=dev); ub = torch.empty(nb.value, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.vf_wino_pack_weights(w.data_ptr(), uf.data_ptr(), ub.data_ptr(), Cout, Cin, st)
stamps = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
for _ in range(3):
    assert lib.vf_wino_conv_fwd(x.data_ptr(), uf.data_ptr(), None, stamps.data_ptr(), None, y.data_ptr(), None, 0, S, Cin, Cout, H, H, 0, st) == 0
torch.cuda.synchronize()
a = stamps.cpu().numpy().reshape(-1, 8); a = a[a[:, 2] != 0]; n = len(a)
wall = (a[:, 7].max() - a[:, 6].min()) / 100e6
life = (a[:, 7] - a[:, 6]) / 100e6
tot = (a[:, 2] - a[:, 0]).astype(float)
flops = 2.0 * S * Cout * Cin * 9 * H * H
print(f"Cin={Cin} Cout={Cout} H={H}: {n} WGs, wall {wall*1e6:.1f} us -> {flops/wall/1e12:.1f} TF(direct-equivalent); WG life {life.mean()*1e6:.1f} us; clock {np.median(tot/np.maximum(life,1e-9))/1e9:.2f} GHz")
print(f"  cycles/WG: total {tot.mean():.0f} | loop {(a[:,1]-a[:,0]).mean():.0f} = store {a[:,3].mean():.0f} + transform {a[:,4].mean():.0f} + mfma {a[:,5].mean():.0f} | epilogue {(a[:,2]-a[:,1]).mean():.0f}; chunks {((Cin+7)//8)}")
