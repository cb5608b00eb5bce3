import os, sys, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from view_fusion_amd import ops
dev = torch.device("cuda:0")
def t_us(fn, n=100):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ops.st.SMALL_CONV_MAX_WGS, ops.st.SMALL_CONV_MAX_CIN3, ops.st.SMALL_CONV_MAX_S3 = (1 << 30, 1 << 30), 1 << 30, 1 << 30
with torch.no_grad():
    for H, Cout in ((32, 128), (16, 192), (64, 64)):
        for Cin in (32, 64, 128, 256, 512):
            conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1).to(dev)
            x = torch.randn(1, Cin, H, H, device=dev)
            print(f"{Cin:4d} -> {Cout:3d} @ {H}x{H} ({Cin // 8 // 8 + (1 if (Cin // 8) % 8 else 0)} rounds): {t_us(lambda: ops.conv2d(x, conv)):6.2f} us", flush=True)
