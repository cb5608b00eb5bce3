"""Isolate the cost of the Winograd forward epilogue operands (bias / per-view bias / residual)."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import _lib, ops
dev = torch.device("cuda:0")
S = 96
lib = _lib.load()
st = ops._stream()
for Cin, Cout, H in [(64, 64, 64), (128, 128, 32), (192, 192, 16), (320, 320, 8)]:
    x = torch.randn(S, Cin, H, H, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / 10
    nf, nb = ctypes.c_long(0), ctypes.c_long(0)
    _lib.call("vf_wino_pack_sizes", Cout, Cin, ctypes.byref(nf), ctypes.byref(nb))
    uf = torch.empty(nf.value, device=dev)
    _lib.call("vf_wino_pack_weights", w.data_ptr(), uf.data_ptr(), None, Cout, Cin, st)
    y = torch.empty(S, Cout, H, H, device=dev)
    bias = torch.randn(Cout, device=dev); vb = torch.randn(S, Cout, device=dev); res = torch.randn_like(y)
    nws = lib.vf_wino_conv_ws_floats(S, Cin, Cout, H, H)
    ws = torch.empty(max(nws, 1), device=dev)
    out = []
    for name, b, v, r in [("none", None, None, None), ("bias+vbias", bias, vb, None), ("bias+res", bias, None, res)]:
        def fn():
            _lib.call("vf_wino_conv_fwd", x.data_ptr(), uf.data_ptr(), b.data_ptr() if b is not None else None,
                      v.data_ptr() if v is not None else None, r.data_ptr() if r is not None else None, y.data_ptr(),
                      ws.data_ptr(), nws, S, Cin, Cout, H, H, 0, st)
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(f"{name} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
    print(f"Cin {Cin} Cout {Cout} H {H}: " + " | ".join(out), flush=True)
