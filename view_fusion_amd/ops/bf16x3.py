"""EXPERIMENT (VF_BF16X3=1, default off): 1x1 convolutions as bf16x3 split products (csrc/conv1x1_bf16x3.hip)."""
import ctypes

import torch

from .. import _lib
from .state import st
from .core import _call, _check, _ptr, _stream


# EXPERIMENT, default off: VF_BF16X3=1 routes the forward and dgrad passes of the 1x1 convolutions (maps >= 8x8) through
# the bf16x3 split-product kernel (csrc/conv1x1_bf16x3.hip); weight gradients stay on the fp32 MFMA kernels.


def _packed_b3(layer, force):
    """Split + packed operands (forward, dgrad) of a 1x1 layer for the bf16x3 kernel; same caching rules as _packed."""
    w = layer.weight
    cache = getattr(layer, "_vf_pack3", None)
    key = (w._version, w.data_ptr(), w.device)
    if not force and cache is not None and cache[0] == key:
        return cache[1], cache[2]
    Cout, Cin = w.shape[0], w.shape[1]
    lib = _lib.load()
    nf, nb = lib.vf_conv1x1_bf16x3_pack_dwords(Cout, Cin), lib.vf_conv1x1_bf16x3_pack_dwords(Cin, Cout)
    if cache is not None and cache[1].numel() == nf and cache[1].device == w.device:
        wf, wb = cache[1], cache[2]
    else:
        wf = torch.empty(nf, device=w.device, dtype=torch.int32)
        wb = torch.empty(nb, device=w.device, dtype=torch.int32)
    wd = w.detach()
    _check(wd)
    _call("vf_conv1x1_bf16x3_pack", _ptr(wd), ctypes.c_void_p(wf.data_ptr()), ctypes.c_void_p(wb.data_ptr()), Cout, Cin,
              _stream())
    object.__setattr__(layer, "_vf_pack3", (None if force else key, wf, wb))
    return wf, wb


def _use_b3(KS, m, HW):
    return st.BF16X3 and KS == 1 and m == 0 and HW >= 64 and (HW & (HW - 1)) == 0
