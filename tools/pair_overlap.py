"""Would the dgrad and the wgrad launch of one 3x3 layer gain from being sibling nodes of the captured iteration?

Both are persistent one-workgroup-per-CU kernels; in the captured iteration they follow each other on one stream.  This
tool replays, as HIP graphs, ten (forward-kernel, weight-gradient) pairs of one layer shape (a) back to back on one stream
and (b) forked onto two streams and joined after every pair, and prints both times per pair.

    python tools/pair_overlap.py
"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import _lib, ops
dev = torch.device("cuda:0")
S, REPS = 96, 10
lib = _lib.load()


def bench(Cin, Cout, H):
    x = torch.randn(S, Cin, H, H, device=dev); dy = torch.randn(S, Cout, H, H, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / 10
    dw = torch.empty(Cout, Cin, 3, 3, device=dev)
    wws = torch.empty(lib.vf_wino_wgrad_ws_floats(S, Cin, Cout, H, H), device=dev)
    f44 = H >= 32
    if f44:
        nf, nb = ctypes.c_long(0), ctypes.c_long(0)
        _lib.call("vf_wino44_pack_sizes", Cout, Cin, ctypes.byref(nf), ctypes.byref(nb))
        uf = torch.empty(nf.value, device=dev)
        _lib.call("vf_wino44_pack_weights", w.data_ptr(), uf.data_ptr(), None, Cout, Cin, ops._stream())
        nws = lib.vf_wino44_conv_ws_floats(S, Cin, Cout, H, H)
    else:
        nf, nb = ctypes.c_long(0), ctypes.c_long(0)
        _lib.call("vf_wino_pack_sizes", Cout, Cin, ctypes.byref(nf), ctypes.byref(nb))
        uf = torch.empty(nf.value, device=dev)
        _lib.call("vf_wino_pack_weights", w.data_ptr(), uf.data_ptr(), None, Cout, Cin, ops._stream())
        nws = lib.vf_wino_conv_ws_floats(S, Cin, Cout, H, H)
    cws = torch.empty(max(nws, 1), device=dev)
    y = torch.empty(S, Cout, H, H, device=dev)

    def conv():
        _lib.call("vf_wino44_conv_fwd" if f44 else "vf_wino_conv_fwd", x.data_ptr(), uf.data_ptr(), None, None, None,
                  y.data_ptr(), cws.data_ptr(), nws, S, Cin, Cout, H, H, 0, ops._stream())

    def wgrad():
        _lib.call("vf_wino_wgrad", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, None, wws.data_ptr(), wws.numel(), S,
                  Cin, Cout, H, H, 0, ops._stream())

    side = torch.cuda.Stream()
    out = []
    for fork in (False, True):
        for _ in range(2):
            conv(); wgrad()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="relaxed"):
            main = torch.cuda.current_stream()
            for _ in range(REPS):
                if fork:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        wgrad()
                    conv()
                    main.wait_stream(side)
                else:
                    conv(); wgrad()
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 5 / REPS * 1e3)
    print(f"{Cin:3d} -> {Cout:3d} @ {H:2d}^2 ({'F(4x4)' if f44 else 'nested'} forward kernel + F(4x4) weight gradient): one stream "
          f"{out[0]:7.1f} us per pair, forked {out[1]:7.1f} us per pair ({out[1] / out[0]:.3f})", flush=True)


for shp in [(64, 64, 64), (128, 64, 64), (128, 128, 32), (256, 128, 32), (192, 192, 16), (384, 192, 16), (320, 320, 8), (640, 320, 8)]:
    bench(*shp)
