import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.chdir(ROOT)
from conftest import SMALL
from view_fusion_amd import UNet, ops
from view_fusion_amd.utils import deterministic_fill_
dev = torch.device('cuda:0')
g = np.load('tests/golden/unet_small.npz')
net = UNet(**SMALL)
deterministic_fill_(net.state_dict())
net = net.to(dev).eval()
for label, f24, f44 in (("natural policy", False, False), ("every eligible layer on the nested F(2,3)xF(4,3) kernel", True, False),
                        ("64x64 / 32x32 layers on the F(4x4,3x3) kernel, nested below", True, True)):
    ops.st.FORCE_WINOGRAD, ops.st.FORCE_WINOGRAD44 = f24, f44
    with torch.no_grad():
        y = net(torch.from_numpy(g['x']).to(dev), torch.from_numpy(g['angle']).to(dev), torch.from_numpy(g['level']).to(dev))
    ref = g['y'] if 'y' in g else g['out']
    err = np.abs(y.cpu().numpy() - ref)
    print(label, '| max abs err', err.max(), 'ref absmax', np.abs(ref).max(), 'rms err', np.sqrt((err**2).mean()))
