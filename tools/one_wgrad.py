"""One Winograd weight-gradient shape, repeated -- target for rocprofv3 --pmc.  usage: one_wgrad.py Cin Cout H [S] [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import _lib, ops
Cin, Cout, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
S = int(sys.argv[4]) if len(sys.argv) > 4 else 96
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda:0")
lib = _lib.load()
x = torch.randn(S, Cin, H, H, device=dev); dy = torch.randn(S, Cout, H, H, device=dev)
dw = torch.empty(Cout, Cin, 3, 3, device=dev)
ws = torch.empty(lib.vf_wino_wgrad_ws_floats(S, Cin, Cout, H, H), device=dev)
st = ops._stream()
for _ in range(reps):
    _lib.call("vf_wino_wgrad", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, None, ws.data_ptr(), ws.numel(), S, Cin, Cout, H, H, 0, st)
torch.cuda.synchronize()
print("done")
