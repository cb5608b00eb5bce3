"""What a captured training iteration needs made current BEFORE the capture starts."""
import torch

from .state import st
from .dense import _ta_desc
from .packing import pack_all


def prime_tables(net, S, device):
    """The per-geometry descriptor tables of a UNet's training forward (weight packs, FeatureWiseAffine group) hold
    ONE geometry at a time and are rebuilt -- with a host-to-device copy -- when S changes which layers take the Winograd
    path: make them current for S now, so that a capture of the iteration that follows finds them and only launches.
    Returns those tables and the packed-weight buffers: the captured launches address them, and the caches drop them
    when another geometry comes along, so the graph's owner keeps them referenced."""
    if not (isinstance(net, torch.nn.Module) and hasattr(net, "_affine_layers")):
        return None
    pack_all(net, S)
    layers = net._affine_layers()
    _ta_desc(layers, S, device)
    return (getattr(net, "_vf_pack_plan", None), st._TA_DESC.get(id(layers[0])),
            [(getattr(m, "_vf_pack", None), getattr(m, "_vf_wpack", None), getattr(m, "_vf_w4pack", None)) for m in net.modules()
             if isinstance(m, torch.nn.Conv2d)])
