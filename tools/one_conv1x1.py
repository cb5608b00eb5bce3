"""One 1x1 conv shape, forward, repeated (target for rocprofv3 --pmc): one_conv1x1.py Cin Cout H [S] [reps]"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import _lib, ops
Cin, Cout, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
S = int(sys.argv[4]) if len(sys.argv) > 4 else 96
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda:0")
st = ops._stream()
x = torch.randn(S, Cin, H, H, device=dev)
w = torch.randn(Cout, Cin, 1, 1, device=dev) / 10
nf, nb = ctypes.c_long(0), ctypes.c_long(0)
_lib.call("vf_conv_pack_sizes", Cout, Cin, 1, ctypes.byref(nf), ctypes.byref(nb))
wf = torch.empty(nf.value, device=dev)
_lib.call("vf_conv_pack_weights", w.data_ptr(), wf.data_ptr(), None, Cout, Cin, 1, st)
y = torch.empty(S, Cout, H, H, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(reps + 3):
    if it == 3:
        e0.record()
    _lib.call("vf_conv_fwd", x.data_ptr(), wf.data_ptr(), None, None, None, y.data_ptr(), None, 0, S, Cin, Cout, H, H, 1, 0, st)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps * 1e3
print(f"{Cin}->{Cout}@{H} S={S}: {t:.1f} us  {2.0 * S * H * H * Cin * Cout / t / 1e6:.1f} TF")
