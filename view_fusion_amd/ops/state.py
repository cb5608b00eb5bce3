"""Every piece of per-process mutable state of the op layer, and every knob, on ONE object (`st`).

The package runs one process per GPU (torchrun), with the forward pass on the main thread and the backward pass on
the autograd worker thread of the same process: both see this object.  Two models or two trainers in one process
share it too -- the deferred-sum lists are tagged with the autograd graph task they belong to and entries of another
task are flushed, not mixed (ops/deferred.py), the caches are keyed on what they cache, and the gradient arena is
`reducer.ACTIVE` (one per process).  Tests and tuning tools set the upper-case knobs (`ops.st.FORCE_WINOGRAD = True`);
the environment variables are read once, here."""
import os


class State:
    def __init__(self):
        # ---- Diagnostics
        self.KERNEL_LOG = None
        self._KIND_OVERRIDE = None       # (attention backward labels its batched GEMMs)
        # ---- Kernel-choice policy (ops/policy.py reads these; tests and tuning tools set them)
        self.WINOGRAD = True           # fused Winograd F(2x2,3x3) for stride-1 3x3 convs on large maps
        self.FORCE_WINOGRAD = False    # tests: take the Winograd path even when the grid would not fill the chip
        self.WINO_MIN_TILES = int(os.environ.get("VF_WINO_MIN_TILES", 30))   # policy thresholds (tuning aid)
        self.WINO_MIN_FILL = int(os.environ.get("VF_WINO_MIN_FILL", 65))
        self.WINO_WGRAD_MIN_TILES = int(os.environ.get("VF_WINO_WGRAD_MIN_TILES", 256))   # measured at B = 4 / 8 (S = 24 / 48)
        self.WINOGRAD_WGRAD = True     # weight gradients of those layers (plain stride-1 ones) through the same transform
        self.WINOGRAD44 = os.environ.get("VF_WINO44", "1") == "1"     # F(4x4,3x3) forward / dgrad kernel on the large maps
        self.FORCE_WINOGRAD44 = False  # tests: F(4x4) wherever the kernel supports the map
        self.SMALL_CONV = os.environ.get("VF_SMALL_CONV", "1") == "1"
        self.SMALL_CONV_MAX_WGS = (int(os.environ.get("VF_SMALL_CONV_MAX_WGS1", 1024)),      # 1x1 layers
                          int(os.environ.get("VF_SMALL_CONV_MAX_WGS3", 512)))       # 3x3 layers
        self.SMALL_CONV_MAX_CIN3 = int(os.environ.get("VF_SMALL_CONV_MAX_CIN3", 256))
        self.SMALL_CONV_MAX_S3 = int(os.environ.get("VF_SMALL_CONV_MAX_S3", 1))
        self.SMALL_PACK = os.environ.get("VF_SMALL_PACK", "1") == "1"
        self.RES_FOLD = os.environ.get("VF_RES_FOLD", "1") == "1"
        self.WINO_GN_FUSION = os.environ.get("VF_WINO_GN", "1") == "1"
        self.ATTN_DSCORE = os.environ.get("VF_ATTN_DSCORE", "1") != "0"
        self.ATTN_DVDK = os.environ.get("VF_ATTN_DVDK", "1") != "0"
        self.BF16X3 = os.environ.get("VF_BF16X3") == "1"
        # ---- Deferred sums of the running backward pass (ops/deferred.py)
        self.ROWSUM_FUSION = True
        self.COLSUM_DEFER = True
        self.WRED_DEFER = os.environ.get("VF_WRED_DEFER", "1") != "0"
        self.WRED_DEFER_GENERIC = os.environ.get("VF_WRED_DEFER_GENERIC", "1") != "0"   # the direct / 1x1 kernels' slabs too (round 6)
        self._PENDING_COLSUMS = []
        self._PENDING_TASK = None      # torch._C._current_graph_task_id() of the backward pass the pending entries belong to
        self._PENDING_WRED = []        # [(row: list of 9 int64, workgroups, keep-alive tensors)]
        self._CAPTURE_TABLE = None   # [device table (rows x 6 int64), rows used, host rows, keep-alive] while a Trainer is capturing
        self._CAPTURE_TABLE_W = None  # the same for the deferred slab sums: [device table (rows x 9 int64), rows used, host rows, keep-alive]
        self._WRED_ARENA = {}          # device -> [arena tensor or None, floats handed out in the running pass]
        self._CS_TABLE = {}          # device -> {"rows": last uploaded table, "ring": [[pinned, device table, event, rows], ...], "next": i}
        self._WR_TABLE = {}          # device -> {"ring": [[pinned, device table, event, key, rows, workgroups], ...], "next": i}
        # ---- Caches
        self._ws = {}
        self._WINO_KIND_CACHE = {}
        self._TA_DESC = {}
        self._TA_GDST = {}
        self._VC_CACHE = []            # [(device view_count tensor, version, (off, S, maxV))], most recent first, bounded
        # tensors the last flush launches still read (kept until the next flush)
        self.keep_wred = None
        self.keep_colsums = None


st = State()
