"""Time the direct and the Winograd weight-gradient kernels on the small UNet's 3x3 stride-1 layer shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import _lib, ops
S = int(os.environ.get("S", 96))
MODE = int(os.environ.get("MODE", 0))      # 2: x stored at half size, nearest-upsampled on read
dev = torch.device("cuda:0")
shapes = [(6, 64, 64), (64, 6, 64), (64, 64, 64), (128, 64, 64), (192, 64, 64), (64, 128, 32), (128, 128, 32),
          (256, 128, 32), (320, 128, 32), (128, 192, 16), (192, 192, 16), (384, 192, 16), (512, 192, 16), (192, 320, 8), (320, 320, 8), (640, 320, 8)]
lib = _lib.load()
st = ops._stream()
for Cin, Cout, H in shapes:
    Hx = H // 2 if MODE == 2 else H
    x = torch.randn(S, Cin, Hx, Hx, device=dev)
    dy = torch.randn(S, Cout, H, H, device=dev)
    dw0 = torch.empty(Cout, Cin, 3, 3, device=dev)
    dw1 = torch.empty_like(dw0)
    db1 = torch.empty(Cout, device=dev)
    n0 = lib.vf_conv_wgrad_ws_floats(S, Cin, Cout, H, H, 3)
    n1 = lib.vf_wino_wgrad_ws_floats(S, Cin, Cout, H, H)
    ws = torch.empty(max(n0, n1), device=dev)
    def direct():
        _lib.call("vf_conv_wgrad", x.data_ptr(), dy.data_ptr(), dw0.data_ptr(), ws.data_ptr(), ws.numel(), S, Cin, Cout,
                  H, H, 3, MODE, st)
    def wino():
        _lib.call("vf_wino_wgrad", x.data_ptr(), dy.data_ptr(), dw1.data_ptr(), db1.data_ptr(), None, ws.data_ptr(), ws.numel(), S, Cin, Cout,
                  H, H, MODE, st)
    res = []
    for fn in (direct, wino):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * S * H * H * Cin * Cout * 9
    err = ((dw1 - dw0).norm() / dw0.norm()).item()
    dbr = dy.sum(dim=(0, 2, 3))
    berr = ((db1 - dbr).norm() / dbr.norm()).item()
    print(f"Cin {Cin:4d} Cout {Cout:4d} H {H:3d}: direct {res[0]*1e3:7.1f} us {fl/res[0]/1e9:6.1f} TF | "
          f"wino {res[1]*1e3:7.1f} us {fl/res[1]/1e9:6.1f} TF | rel diff {err:.2e} bias {berr:.1e}", flush=True)
