"""Where a chunk of the Winograd F(4x4) weight-gradient kernel spends its cycles, per role (dM waves / V waves), from a
-DVF_STAMPS44 build of winograd44.hip (shader-clock cycles, s_memtime):

    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -DVF_STAMPS44 -c view_fusion_amd/csrc/winograd44.hip -o build/ab/w44_st.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libvf_STAMPS.so <build/vf_hip/*.o except winograd44.o> build/ab/w44_st.o
    VF_DEBUG_AB=1 VF_HIP_LIB=$PWD/build/ab/libvf_STAMPS.so python tools/wgrad_stamps.py
"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import _lib, ops
dev = torch.device("cuda:0")
S = 96
lib = _lib.load(); st = ops._stream()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.vf_debug_stamps44.argtypes = [ctypes.c_void_p, ctypes.c_int]
for Cin, Cout, H in [(64, 64, 64), (192, 64, 64), (320, 128, 32), (512, 192, 16), (320, 320, 8)]:
    x = torch.randn(S, Cin, H, H, device=dev); dy = torch.randn(S, Cout, H, H, device=dev)
    dw = torch.empty(Cout, Cin, 3, 3, device=dev)
    ws = torch.empty(lib.vf_wino_wgrad_ws_floats(S, Cin, Cout, H, H), device=dev)
    f = lambda: _lib.call("vf_wino_wgrad", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, None, ws.data_ptr(), ws.numel(), S, Cin, Cout, H, H, 0, st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    raw.vf_debug_stamps44(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    raw.vf_debug_stamps44(out, 1)
    v = [float(t) for t in out]
    line = f"{Cin}->{Cout}@{H}: {e0.elapsed_time(e1) * 100:.1f} us/launch"
    for role, name in ((0, "dM"), (1, "V ")):
        a, b, c, n, pro, epi, waves = v[16 * role: 16 * role + 7]
        ph = [t / waves for t in v[16 * role + 8: 16 * role + 14]]
        line += (f" | {name}: chunk {(a + b + c) / n:6.0f} cyc = to-barrier {a / n:6.0f} + in-barrier {b / n:5.0f} + after {c / n:5.0f};"
                 f" chunks/wave {n / waves:5.1f}, prologue {pro / waves:6.0f}, epilogue {epi / waves:6.0f}"
                 f" [G {ph[0]:.0f} | sync {ph[1]:.0f} | write+sync {ph[2]:.0f} | read {ph[3]:.0f} | sync+write+sync {ph[4]:.0f} | store {ph[5]:.0f}]")
    line += f" | in-kernel clock {v[14] / v[15] * 0.1:.2f} GHz, wave lifetime {v[15] / v[6] / 100:.1f} us (dM) {v[31] / v[22] / 100:.1f} us (V)"
    print(line, flush=True)
