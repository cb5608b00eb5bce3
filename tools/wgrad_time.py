"""us per launch of vf_wino_wgrad on a few shapes (timing only; used with ablation builds via VF_DEBUG_AB / VF_HIP_LIB)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import _lib, ops
S = 96
dev = torch.device("cuda:0")
lib = _lib.load(); st = ops._stream()
out = []
for Cin, Cout, H in [(64, 64, 64), (192, 64, 64), (320, 128, 32), (512, 192, 16), (320, 320, 8)]:
    x = torch.randn(S, Cin, H, H, device=dev); dy = torch.randn(S, Cout, H, H, device=dev)
    dw = torch.empty(Cout, Cin, 3, 3, device=dev)
    ws = torch.empty(lib.vf_wino_wgrad_ws_floats(S, Cin, Cout, H, H), device=dev)
    f = lambda: _lib.call("vf_wino_wgrad", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, None, ws.data_ptr(), ws.numel(), S, Cin, Cout, H, H, 0, st)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    out.append(f"{Cin}->{Cout}@{H}: {e0.elapsed_time(e1) * 100:.1f}")
print(os.environ.get("VF_HIP_LIB", "default"), " | ".join(out))
