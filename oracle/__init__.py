"""CPU oracle for the ViewFusion hot path (TEST INFRASTRUCTURE ONLY).

This package is a fresh fp32 PyTorch-CPU restatement of the reference's
`model/unet.py` and `model/view_fusion.py` arithmetic.  It is the checker the
HIP path is compared against; it is never the thing measured or shipped.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import it.  The product package (`view_fusion_amd`) must not.

Parity pin: the reference repository has no tests / golden vectors of its own
("parity unpinned" by the reference).  The oracle is therefore pinned against
outputs of the real reference imported on CPU in the build container; the
vectors and the generating script live in `tests/golden/`.
"""
