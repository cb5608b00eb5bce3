"""Launch plumbing of the op layer: raw pointers, the current stream, argument checks, the C-ABI call wrappers
(with the optional kernel log that bench.py / the tools read) and the shared split-K workspace."""
import ctypes

import torch

from .. import _lib
from .state import st


_MODES = {"same": 0, "down2": 1, "up2": 2}


def _ptr(t, offset_elems=0):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr() + 4 * offset_elems)


def _raw_stream():
    # torch.cuda.current_stream() builds a Python Stream object (~5 us); the raw handle is all a launcher needs
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _stream():
    return ctypes.c_void_p(_raw_stream())


def _check(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.VFHipError("view_fusion_amd ops need CUDA/HIP tensors (no CPU fallback)")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.VFHipError(f"expected contiguous float32, got {t.dtype} contiguous={t.is_contiguous()}")


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# Optional per-launch timing (bench.py roofline leg).  When st.KERNEL_LOG is a list, every launch that goes through
# _launch is bracketed by HIP events recorded on the stream the kernel is launched on, and
# (kind, algorithmic flops, start, end, tag, C-ABI entry point, algorithmic HBM bytes) is appended.


def _launch(kind, flops, name, *args, tag=None, nbytes=0.0):
    if st.KERNEL_LOG is None:
        _lib.call(name, *args)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.call(name, *args)
    e1.record()
    st.KERNEL_LOG.append((kind, flops, e0, e1, tag, name, nbytes))


# Launchers without a roofline of their own still get a family in the bench's table (so that the table sums to the
# instrumented step): kind by entry point, "misc" otherwise.
_CALL_KIND = {"vf_wino44_pack_weights": "pack", "vf_wino44_pack_weights_multi": "pack", "vf_bgemm": "bgemm", "vf_softmax_bwd": "attn_bwd", "vf_softmax_fwd": "attn_fwd",
              "vf_colsum": "reduce", "vf_colsum_multi": "reduce", "vf_rowsum": "reduce", "vf_bias_grad": "reduce",
              "vf_sumpool2": "reduce", "vf_conv_pack_weights": "pack", "vf_wino_pack_weights": "pack",
              "vf_conv_pack_weights_multi": "pack", "vf_wino_pack_weights_multi": "pack",
              "vf_time_affine_fwd": "embed", "vf_time_affine_bwd": "embed", "vf_sincos_embed": "embed",
              "vf_swish_fwd": "embed", "vf_swish_bwd": "embed",
              "vf_stack_views": "diffusion", "vf_compose_fwd": "diffusion", "vf_compose_mse_bwd": "diffusion",
              "vf_gather_level": "diffusion", "vf_p_sample_tail": "diffusion"}


def _call(name, *args, flops=0.0, nbytes=0.0):
    if st.KERNEL_LOG is None:
        _lib.call(name, *args)
        return
    _launch(st._KIND_OVERRIDE or _CALL_KIND.get(name, "misc"), flops, name, *args, nbytes=nbytes)


# ---------------------------------------------------------------------------------------------
# split-K / slab workspace: one buffer per (device, stream) -- launches on one stream are ordered, so consecutive
# kernels may reuse it; two streams (two models driven concurrently) get separate buffers.  Grown on demand.
# Bounded: at most _WS_MAX (device, stream) entries, least recently used evicted (generate() makes a side stream per
# graph warm-up).  A buffer allocated WHILE a stream is capturing lives in that graph's private pool: it is handed to
# the capture but never cached, so no later capture or eager launch can pick up memory owned by another graph.
_WS_MAX = 4


def _workspace(device, nfloats):
    key = (device, _raw_stream())
    buf = st._ws.pop(key, None)
    if buf is None or buf.numel() < nfloats:
        new = torch.empty(int(nfloats), device=device, dtype=torch.float32)
        if torch.cuda.is_current_stream_capturing():
            if buf is not None:
                st._ws[key] = buf
            return new
        buf = new
    st._ws[key] = buf                                   # (re-)inserted last = most recently used
    while len(st._ws) > _WS_MAX:
        st._ws.pop(next(iter(st._ws)))
    return buf
