import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import ops
Cin, Cout, H = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
layer = torch.nn.Conv2d(Cin, Cout, 3, padding=1).to(dev)
x = torch.rand(96, Cin, H, H, device=dev)
with torch.no_grad():
    for _ in range(6):
        y = ops.conv2d(x, layer)
torch.cuda.synchronize()
