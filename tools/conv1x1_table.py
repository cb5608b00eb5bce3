"""Time vf_conv_fwd on the small UNet's 1x1 shapes (forward roles and the swapped dgrad roles).
Env VF_CONV_NCO / VF_CONV_NPT force the channel-tile multiplier / pixel-tile size (tuning aids)."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import _lib, ops
dev = torch.device("cuda:0")
S = 96
st = ops._stream()
shapes = [(192, 576, 16), (576, 192, 16), (192, 192, 16), (384, 192, 16), (192, 384, 16), (512, 192, 16), (128, 64, 64),
          (64, 128, 64), (192, 64, 64), (256, 128, 32), (128, 256, 32), (320, 128, 32), (640, 320, 8), (320, 640, 8),
          (320, 960, 8), (960, 320, 8), (320, 320, 8)]
tot = 0.0
tot3 = 0.0
for Cin, Cout, H in shapes:
    x = torch.randn(S, Cin, H, H, device=dev)
    w = torch.randn(Cout, Cin, 1, 1, device=dev) / 10
    nf, nb = ctypes.c_long(0), ctypes.c_long(0)
    _lib.call("vf_conv_pack_sizes", Cout, Cin, 1, ctypes.byref(nf), ctypes.byref(nb))
    wf = torch.empty(nf.value, device=dev)
    _lib.call("vf_conv_pack_weights", w.data_ptr(), wf.data_ptr(), None, Cout, Cin, 1, st)
    y = torch.empty(S, Cout, H, H, device=dev)
    bias = torch.randn(Cout, device=dev)
    def fn():
        _lib.call("vf_conv_fwd", x.data_ptr(), wf.data_ptr(), bias.data_ptr(), None, None, y.data_ptr(), None, 0, S, Cin,
                  Cout, H, H, 1, 0, st)
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    ref = torch.nn.functional.conv2d(x[:2].double(), w.double(), bias.double())        # fp64 reference
    err = ((y[:2].double() - ref).norm() / ref.norm()).item()
    # the bf16x3 experiment on the same operands (csrc/conv1x1_bf16x3.hip)
    lib = _lib.load()
    w3 = torch.empty(lib.vf_conv1x1_bf16x3_pack_dwords(Cout, Cin), device=dev, dtype=torch.int32)
    _lib.call("vf_conv1x1_bf16x3_pack", w.data_ptr(), w3.data_ptr(), None, Cout, Cin, st)
    y3 = torch.empty_like(y)
    def fn3():
        _lib.call("vf_conv1x1_bf16x3", x.data_ptr(), None, 0, w3.data_ptr(), bias.data_ptr(), None, None, y3.data_ptr(), None, 0,
                  S, Cin, Cout, H * H, st)
    for _ in range(3): fn3()
    e0.record()
    for _ in range(20): fn3()
    e1.record(); torch.cuda.synchronize()
    t3 = e0.elapsed_time(e1) / 20 * 1e3
    err3 = ((y3[:2].double() - ref).norm() / ref.norm()).item()
    tot3 += t3
    fl = 2.0 * S * H * H * Cin * Cout
    mb = 4.0 * S * H * H * (Cin + Cout) / 1e6
    tot += t
    print(f"{Cin:4d}->{Cout:4d} @{H:2d}: fp32 MFMA {t:7.1f} us {fl / t / 1e6:6.1f} TF  {mb / t:5.2f} TB/s  rel-L2 err vs fp64 {err:.1e}"
          f" | bf16x3 {t3:7.1f} us {fl / t3 / 1e6:6.1f} TF (fp32-equivalent)  err {err3:.1e}", flush=True)
print(f"total fp32 MFMA {tot:.0f} us, bf16x3 {tot3:.0f} us")
