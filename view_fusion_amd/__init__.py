"""view_fusion_amd -- MI355X-native (gfx950) engine for the ViewFusion hot path.

    from view_fusion_amd import UNet, ViewFusion

Drop-in for the reference's `model.unet.UNet` / `model.view_fusion.ViewFusion` (same
constructor / forward signatures and state_dict).  The arithmetic runs in hand-written HIP
kernels (libvf_hip.so, C ABI in include/vf_hip.h); importing the package works anywhere, but
running it needs the built library and a GPU -- there is no CPU fallback.
"""
from .unet import UNet
from .view_fusion import ViewFusion

__all__ = ["UNet", "ViewFusion"]
