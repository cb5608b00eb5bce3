"""Where a Winograd forward tile's time goes: prologue / chunk loop / epilogue, in shader-clock cycles (s_memtime), from a -DVF_STAMPS build of winograd24.hip:

    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -DVF_STAMPS -c view_fusion_amd/csrc/winograd24.hip -o w24s.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o view_fusion_amd/lib/libvf_stamps.so build/vf_hip/<all but winograd24>.o w24s.o
    VF_HIP_LIB=$PWD/view_fusion_amd/lib/libvf_stamps.so python tools/wino_stamps.py
"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import _lib, ops
dev = torch.device("cuda:0")
S = 96
lib = _lib.load()
st = ops._stream()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.vf_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for Cin, Cout, H in [(64, 64, 64), (128, 64, 64), (128, 128, 32), (256, 128, 32), (192, 192, 16), (320, 320, 8)]:
    x = torch.randn(S, Cin, H, H, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / 10
    nf, nb = ctypes.c_long(0), ctypes.c_long(0)
    _lib.call("vf_wino_pack_sizes", Cout, Cin, ctypes.byref(nf), ctypes.byref(nb))
    uf = torch.empty(nf.value, device=dev)
    _lib.call("vf_wino_pack_weights", w.data_ptr(), uf.data_ptr(), None, Cout, Cin, st)
    y = torch.empty(S, Cout, H, H, device=dev)
    nws = lib.vf_wino_conv_ws_floats(S, Cin, Cout, H, H)
    ws = torch.empty(max(nws, 1), device=dev)
    ops_on = os.environ.get("VF_STAMP_OPERANDS", "0") == "1"     # 1: with bias + per-view bias
    bias = torch.randn(Cout, device=dev); vb = torch.randn(S, Cout, device=dev)
    def fn():
        _lib.call("vf_wino_conv_fwd", x.data_ptr(), uf.data_ptr(), bias.data_ptr() if ops_on else None,
                  vb.data_ptr() if ops_on else None, None, y.data_ptr(),
                  ws.data_ptr(), nws, S, Cin, Cout, H, H, 0, st)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    raw.vf_debug_stamps(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    raw.vf_debug_stamps(out, 1)
    pro, loop, epi, tiles, ea, eb, ec = [float(v) for v in out][:7]
    nch = (Cin + 7) // 8
    print(f"Cin {Cin:3d} Cout {Cout:3d} H {H:2d}: {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us/launch | whole tiles/launch {tiles / 10:6.0f}"
          f" | cycles per tile: prologue {pro / tiles:6.0f}  loop {loop / tiles:6.0f} ({loop / tiles / nch:5.0f}/chunk)"
          f"  epilogue {epi / tiles:6.0f} = columns+prefetch {ea / tiles:5.0f} + publish+barrier {eb / tiles:5.0f}"
          f" + read+rows {ec / tiles:5.0f} + operands+stores {(epi - ea - eb - ec) / tiles:5.0f}", flush=True)
    x = [float(v) for v in out][8:16]
    if x[6]:
        print("      epilogue tail, cycles: operand loads %5.0f | stage rows of next tile %5.0f | barrier %5.0f | stores %5.0f | stage V(0) %5.0f | barrier %5.0f"
              % tuple(v / x[6] for v in x[:6]), flush=True)
