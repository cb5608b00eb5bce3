import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import _lib, ops
dev = torch.device("cuda:0"); S = 96; st = ops._stream(); lib = _lib.load()
for Cin, Cout, H in [(6, 64, 64), (64, 6, 64)]:
    x = torch.randn(S, Cin, H, H, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) / 10
    y = torch.empty(S, Cout, H, H, device=dev); bias = torch.randn(Cout, device=dev)
    nf, nb = ctypes.c_long(0), ctypes.c_long(0)
    _lib.call("vf_wino_pack_sizes", Cout, Cin, ctypes.byref(nf), ctypes.byref(nb))
    uf = torch.empty(nf.value, device=dev); ub = torch.empty(nb.value, device=dev)
    _lib.call("vf_wino_pack_weights", w.data_ptr(), uf.data_ptr(), ub.data_ptr(), Cout, Cin, st)
    _lib.call("vf_conv_pack_sizes", Cout, Cin, 3, ctypes.byref(nf), ctypes.byref(nb))
    wf = torch.empty(nf.value, device=dev); wb = torch.empty(nb.value, device=dev)
    _lib.call("vf_conv_pack_weights", w.data_ptr(), wf.data_ptr(), wb.data_ptr(), Cout, Cin, 3, st)
    dy = torch.randn_like(y); dx = torch.empty_like(x)
    fns = {
        "wino fwd": lambda: _lib.call("vf_wino_conv_fwd", x.data_ptr(), uf.data_ptr(), bias.data_ptr(), None, None, y.data_ptr(), None, 0, S, Cin, Cout, H, H, 0, st),
        "direct fwd": lambda: _lib.call("vf_conv_fwd", x.data_ptr(), wf.data_ptr(), bias.data_ptr(), None, None, y.data_ptr(), None, 0, S, Cin, Cout, H, H, 3, 0, st),
        "wino dgrad": lambda: _lib.call("vf_wino_conv_fwd", dy.data_ptr(), ub.data_ptr(), None, None, None, dx.data_ptr(), None, 0, S, Cout, Cin, H, H, 0, st),
        "direct dgrad": lambda: _lib.call("vf_conv_fwd", dy.data_ptr(), wb.data_ptr(), None, None, None, dx.data_ptr(), None, 0, S, Cout, Cin, H, H, 3, 0, st),
    }
    out = []
    for name, fn in fns.items():
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(f"{name} {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us")
    print(f"{Cin}->{Cout}@{H}: " + " | ".join(out), flush=True)
