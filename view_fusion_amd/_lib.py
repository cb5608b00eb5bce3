"""ctypes binding of libvf_hip.so (C ABI declared in include/vf_hip.h).

There is NO fallback: if the library is missing or a call returns a non-zero hipError_t the
caller gets an exception.  The library must be loaded after `import torch` so that it binds to
the HIP runtime torch already loaded (same SONAME libamdhip64.so.7) -- one runtime, shared
streams and allocations.
"""
import ctypes
import os

import torch  # noqa: F401  (loads libamdhip64 first)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvf_hip.so")
# Kernel-tuning aid, OFF in production: with VF_DEBUG_AB=1 *and* VF_HIP_LIB=<path> another build of the same C ABI is
# loaded instead (A/B timing of kernel variants in one process tree; such builds live under build/ab/, never in lib/).
# The override is announced on stderr and load() checks every symbol of the table below against it.
_OVERRIDE = os.environ.get("VF_HIP_LIB") if os.environ.get("VF_DEBUG_AB") == "1" else None
if _OVERRIDE:
    LIB_PATH = _OVERRIDE
elif os.environ.get("VF_HIP_LIB"):
    import warnings
    warnings.warn("VF_HIP_LIB is ignored unless VF_DEBUG_AB=1 (kernel-tuning aid)", stacklevel=2)

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float

# name -> argtypes (restype is int unless listed in _RESTYPE)
SIGNATURES = {
    "vf_gn_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "vf_gn_bwd_emits_rowsum": [_I, _I, _I],
    "vf_gn_cat_fwd": [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "vf_gn_cat_bwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vf_conv1x1_cat_fwd": [_P, _P, _I, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P],
    "vf_conv1x1_cat_dgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vf_conv1x1_cat_wgrad": [_P, _P, _I, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P],
    "vf_gn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vf_rowsum": [_P, _P, _I, _I, _P],
    "vf_bias_grad": [_P, _P, _P, _I, _I, _I, _P],
    "vf_colsum": [_P, _P, _I, _I, _I, _P],
    "vf_colsum_multi": [_P, _I, _L, _P],
    "vf_conv_pack_sizes": [_I, _I, _I, ctypes.POINTER(_L), ctypes.POINTER(_L)],
    "vf_conv_pack_weights": [_P, _P, _P, _I, _I, _I, _P],
    "vf_conv_pack_weights_multi": [_P, _I, _L, _P],
    "vf_conv_fwd": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P],
    "vf_conv_fwd_ws_floats": [_I, _I, _I, _I, _I, _I],
    "vf_conv_fwd_gn": [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _F, _I, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P],
    "vf_conv_wgrad_ws_floats": [_I, _I, _I, _I, _I, _I],
    "vf_conv_wgrad": [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P],
    "vf_conv1x1_bf16x3_pack_dwords": [_I, _I],
    "vf_conv1x1_bf16x3_pack": [_P, _P, _P, _I, _I, _P],
    "vf_conv1x1_bf16x3": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vf_wino_supported": [_I, _I, _I],
    "vf_wino_pack_sizes": [_I, _I, ctypes.POINTER(_L), ctypes.POINTER(_L)],
    "vf_wino_pack_weights": [_P, _P, _P, _I, _I, _P],
    "vf_wino_pack_weights_multi": [_P, _I, _L, _P],
    "vf_wino_conv_ws_floats": [_I, _I, _I, _I, _I],
    "vf_wino_conv_fill_pct": [_I, _I, _I, _I, _I, ctypes.POINTER(_I)],
    "vf_wino_conv_fwd": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "vf_wino_conv_gn_fusable": [_I, _I, _I, _I, _I, _I, _I],
    "vf_wino_conv_fwd_gn": [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _F, _I, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "vf_wino44_supported": [_I, _I, _I],
    "vf_wino44_pack_sizes": [_I, _I, ctypes.POINTER(_L), ctypes.POINTER(_L)],
    "vf_wino44_pack_weights": [_P, _P, _P, _I, _I, _P],
    "vf_wino44_pack_weights_multi": [_P, _I, _L, _P],
    "vf_wino44_conv_ws_floats": [_I, _I, _I, _I, _I],
    "vf_wino44_conv_fill_pct": [_I, _I, _I, _I, _I, ctypes.POINTER(_I)],
    "vf_wino44_conv_fwd": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "vf_wino_wgrad_ws_floats": [_I, _I, _I, _I, _I],
    "vf_wino_wgrad_supported": [_I, _I, _I],
    "vf_wino_wgrad": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "vf_wino_wgrad_main": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "vf_wino44_reduce_multi": [_P, _I, _I, _P],
    "vf_conv_wgrad_main": [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "vf_conv1x1_cat_wgrad_main": [_P, _P, _I, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P, _P, _P],
    "vf_sumpool2": [_P, _P, _L, _I, _P],
    "vf_time_affine_fwd": [_P, _I, _P, _P, _I, _I, _I, _P],
    "vf_time_affine_ws_floats": [_I, _I],
    "vf_time_affine_bwd": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vf_bgemm": [_P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _L, _L, _L, _L, _F, _F, _P],
    "vf_attention_fwd": [_P, _P, _P, _I, _I, _I, _P],
    "vf_attention_dscore": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vf_attention_dvdk": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vf_softmax_fwd": [_P, _P, _I, _I, _P],
    "vf_softmax_bwd": [_P, _P, _P, _I, _I, _P],
    "vf_sincos_embed": [_P, _P, _P, _I, _I, _P],
    "vf_swish_fwd": [_P, _P, _L, _P],
    "vf_swish_bwd": [_P, _P, _P, _L, _P],
    "vf_concat_channels": [_P, _P, _P, _I, _L, _L, _I, _P],
    "vf_dropout": [_P, _P, _P, _L, _F, _P],
    "vf_adam_multi": [_P, _I, _L, _F, _F, _F, _F, _F, _F, _P],
    "vf_conv_small_supported": [_I, _I, _I, _I, _I, _I],
    "vf_conv_small": [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "vf_conv_small_res": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P, _P],
    "vf_conv_small_gn": [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _F, _I, _P, _P, _P, _I, _I,
                         _P, _P, _P, _I, _P],
    "vf_conv_small_pack_floats": [_I, _I],
    "vf_conv_small_pack": [_P, _P, _I, _I, _P],
    "vf_adam_multi_dev": [_P, _I, _L, _P, _F, _F, _F, _P],
    "vf_adam_set_scalars": [_P, _F, _F, _F, _P],
    "vf_xgmi_alloc": [ctypes.POINTER(_P), _L],
    "vf_xgmi_free": [_P],
    "vf_xgmi_export": [_P, _P],
    "vf_xgmi_open": [_P, ctypes.POINTER(_P)],
    "vf_xgmi_close": [_P],
    "vf_xgmi_signal": [ctypes.POINTER(_P), _I, _I, _I, ctypes.c_uint, _P],
    "vf_xgmi_wait": [_P, _I, _I, _I, ctypes.c_uint, _P, _L, _P],
    "vf_xgmi_reduce_adam": [_P, _I, _L, ctypes.POINTER(_P), _P, _P, _I, _P, _F, _F, _F, _P],
    "vf_psnr": [_P, _P, _P, _I, _I, _P],
    "vf_gather_level": [_P, _P, _P, _P, _I, _P],
    "vf_stack_views": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vf_compose_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vf_compose_mse_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vf_p_sample_tail": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
}
_RESTYPE = {"vf_conv1x1_bf16x3_pack_dwords": _L, "vf_conv_wgrad_ws_floats": _L, "vf_time_affine_ws_floats": _L, "vf_wino_conv_ws_floats": _L, "vf_wino44_conv_ws_floats": _L, "vf_wino_wgrad_ws_floats": _L, "vf_conv_fwd_ws_floats": _L,
            "vf_conv_small_pack_floats": _L}

_lib = None
N_CALLS = 0          # C-ABI launcher invocations so far (bench.py: launches per sampler step)


class VFHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VFHipError(
                f"{LIB_PATH} not found: build it with `python -m view_fusion_amd.build` "
                "(there is no CPU / eager fallback for the ViewFusion hot path)")
        lib = ctypes.CDLL(LIB_PATH)
        if _OVERRIDE:
            import sys
            print(f"[view_fusion_amd] VF_DEBUG_AB: loading {LIB_PATH} instead of lib/libvf_hip.so", file=sys.stderr)
        missing = [name for name in SIGNATURES if not hasattr(lib, name)]
        if missing:
            raise VFHipError(f"{LIB_PATH} does not export {missing}: stale build? run `python -m view_fusion_amd.build`")
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, _I)
        _lib = lib
    return _lib


def call(name, *args):
    """Invoke an int-returning launcher; non-zero hipError_t -> exception."""
    global N_CALLS
    N_CALLS += 1
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise VFHipError(f"{name} failed with hipError_t {rc}")
