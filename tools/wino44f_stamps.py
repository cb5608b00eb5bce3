"""Where an F(4x4) forward tile's time goes (chunk loop / epilogue phases), in shader-clock cycles (s_memtime), from a
-DVF_STAMPS44F build of winograd44f.hip:

    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -DVF_STAMPS44F -c view_fusion_amd/csrc/winograd44f.hip -o build/ab/w44fs.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libvf_stamps44f.so build/vf_hip/<all but winograd44f>.o build/ab/w44fs.o
    VF_DEBUG_AB=1 VF_HIP_LIB=$PWD/build/ab/libvf_stamps44f.so python tools/wino44f_stamps.py
"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import _lib, ops
dev = torch.device("cuda:0")
S = int(os.environ.get("VF_STAMP_S", "96"))
lib = _lib.load()
st = ops._stream()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.vf_debug_stamps44f.argtypes = [ctypes.c_void_p, ctypes.c_int]
SHAPES = [(64, 64, 64), (128, 64, 64), (192, 64, 64), (128, 128, 32), (256, 128, 32)]
if len(sys.argv) > 1:
    SHAPES = SHAPES[:int(sys.argv[1])]
for Cin, Cout, H in SHAPES:
    x = torch.randn(S, Cin, H, H, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / 10
    nf, nb = ctypes.c_long(0), ctypes.c_long(0)
    _lib.call("vf_wino44_pack_sizes", Cout, Cin, ctypes.byref(nf), ctypes.byref(nb))
    uf = torch.empty(nf.value, device=dev)
    _lib.call("vf_wino44_pack_weights", w.data_ptr(), uf.data_ptr(), None, Cout, Cin, st)
    y = torch.empty(S, Cout, H, H, device=dev)
    nws = lib.vf_wino44_conv_ws_floats(S, Cin, Cout, H, H)
    ws = torch.empty(max(nws, 1), device=dev)
    def fn():
        _lib.call("vf_wino44_conv_fwd", x.data_ptr(), uf.data_ptr(), None, None, None, y.data_ptr(), ws.data_ptr(), nws, S, Cin,
                  Cout, H, H, 0, st)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    raw.vf_debug_stamps44f(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    raw.vf_debug_stamps44f(out, 1)
    v = [float(q) for q in out]
    loop, epi, tiles = v[0], v[1], v[2]
    nch = (Cin + 7) // 8
    print(f"Cin {Cin:3d} Cout {Cout:3d} H {H:2d}: {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us/launch | whole tiles/launch {tiles / 10:6.0f}"
          f" | cycles per tile: loop {loop / tiles:6.0f} ({loop / tiles / nch:5.0f}/chunk, MFMA issue 4608)  epilogue incl. next-tile staging {epi / tiles:6.0f}"
          f" = loads + publish 0 {v[5] / tiles:5.0f} | publish 1 + finish 0 {v[6] / tiles:5.0f} | publish 2 + finish 1 {v[7] / tiles:5.0f}"
          f" | publish 3 + finish 2 + stage rows {v[8] / tiles:5.0f} | finish 3 + V(0) {v[9] / tiles:5.0f} (each up to its barrier)"
          f" | in-kernel clock {v[10] / max(v[11], 1) * 0.1:.2f} GHz", flush=True)
