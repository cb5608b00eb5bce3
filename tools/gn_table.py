"""GroupNorm(+Swish) forward / backward kernels on the small UNet's shapes through the C ABI directly (no autograd,
no host overhead in the timed region): us per launch and algorithmic GB/s (8 B/elem forward, 12 B/elem backward).

    python tools/gn_table.py            # VF_GN_PIPE=0 selects the one-unit-per-workgroup kernels for A/B
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import _lib, ops
S = int(os.environ.get("S", 96))
dev = torch.device("cuda:0")
shapes = [(64, 64), (128, 64), (192, 64), (128, 32), (256, 32), (320, 32), (192, 16), (384, 16), (512, 16), (576, 16), (320, 8), (640, 8)]
st = ops._stream()
P = lambda t: t.data_ptr() if t is not None else None
for C, H in shapes:
    x = torch.randn(S, C, H, H, device=dev); dy = torch.randn_like(x); y = torch.empty_like(x); dx = torch.empty_like(x)
    g = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev)
    mean = torch.empty(S * 32, device=dev); rstd = torch.empty_like(mean)
    parts = torch.empty(2, S, C, device=dev); rows = torch.empty(S, C, device=dev)
    def fwd():
        _lib.call("vf_gn_fwd", P(x), P(g), P(b), P(y), P(mean), P(rstd), S, C, H * H, 32, 1e-5, 1, st)
    def bwd():
        _lib.call("vf_gn_cat_bwd", P(x), None, C, P(g), P(b), P(mean), P(rstd), P(dy), None, None, P(dx), None, P(parts[0]),
                  P(parts[1]), P(rows), S, C, H * H, 32, 1, st)
    res = []
    for fn in (fwd, bwd):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20)
    nb = x.numel() * 4
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(x, 32, g, b, 1e-5))
    err = (y - ref).abs().max().item()
    print(f"C {C:4d} H {H:3d}: fwd {res[0]*1e3:7.1f} us {2*nb/res[0]/1e6:7.0f} GB/s | bwd {res[1]*1e3:7.1f} us {3*nb/res[1]/1e6:7.0f} GB/s | fwd max err {err:.1e}", flush=True)
