"""Host-side op layer: torch.autograd Functions and no-grad entry points over the C ABI of libvf_hip.so.

    state      the one object (`st`) that holds every knob and every piece of per-process mutable state
    policy     which kernel runs a conv layer at a given shape (the kernel-choice policy, in one place)
    core       pointers / streams / C-ABI call wrappers / kernel log
    deferred   gradient destinations, deferred GroupNorm column sums and Winograd slab sums, capture tables
    packing    packed-weight caches, `pack_all`
    norm, conv, dense, attention, diffusion    the ops themselves
    bf16x3     experiment (default off)

There is no CPU / eager fallback: CPU tensors raise VFHipError."""
from .. import _lib, reducer  # noqa: F401
from .state import st  # noqa: F401
from . import state, core, deferred, policy, packing, bf16x3, norm, conv, dense, attention, diffusion, capture  # noqa: F401
from .core import (  # noqa: F401
    _CALL_KIND, _MODES, _WS_MAX, _c, _call, _check, _launch, _ptr, _raw_stream, _stream, _workspace
)
from .deferred import (  # noqa: F401
    _CS_RING, _colsum, _defer_begin, _defer_ok, _flush_colsums, _flush_wred, _gout, _gslot, _rowsum_get,
    _rowsum_put, _wred_rows, _wred_ws, begin_capture, drop_pending_colsums, end_capture, flush_colsums,
    wred_arena_bytes
)
from .capture import (  # noqa: F401
    prime_tables
)
from .policy import (  # noqa: F401
    _WINO_COST, _wino_cycles, _wino_kind, can_fold_residual, use_small_conv, use_winograd, use_winograd_wgrad,
    wino_kind
)
from .packing import (  # noqa: F401
    _WINO_ABI, _conv_ws, _packed, _packed_small, _packed_wino, _wino_ws, pack_all
)
from .bf16x3 import (  # noqa: F401
    _packed_b3, _use_b3
)
from .norm import (  # noqa: F401
    _GroupNormCatSkipFn, _GroupNormFn, _GroupNormSkipFn, _gn_backward, _gn_forward, cat_fusable, group_norm,
    group_norm_cat_skip, group_norm_skip
)
from .conv import (  # noqa: F401
    _Conv1x1CatFn, _Conv2dFn, _conv_small, _conv_small_res, conv1x1_cat, conv2d, conv2d_gn
)
from .dense import (  # noqa: F401
    _DropoutFn, _LinearFn, _SwishFn, _TimeAffineFn, _bgemm, _ta_desc, dropout, linear, sincos_embedding, swish,
    time_affine_all
)
from .attention import (  # noqa: F401
    _AttentionFn, _ConcatFn, attention, concat_channels
)
from .diffusion import (  # noqa: F401
    _ComposeLossFn, compose, compose_mse_loss, gather_level, p_sample_tail, psnr, stack_views, view_offsets
)
