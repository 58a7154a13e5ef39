"""CPU oracle for the Module2 generator / discriminator hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package, and only as the checker /
the CPU baseline.  The product path (``animateportrait_amd``) never imports it
and has no CPU fallback: it fails loudly when the HIP library is missing.

What it is: a restatement, on plain PyTorch CPU fp32 ops, of the algorithms the
reference runs for the hot path.  Each
function cites the reference file:line it follows.  The arithmetic itself lives
in a third-party dependency of the reference (PyTorch, pinned by the reference
to ``torch==1.8.2+cu111``, readme.md:37; this container runs torch 2.10.0 CPU):
conv2d / conv_transpose2d / instance_norm are called through
``torch.nn.functional``; the index-sensitive gathers (grid_sample, bilinear
resize, reflection padding, TPS warp) are additionally restated as explicit
formulas in ``oracle.warp`` / ``oracle.tps`` and cross-checked in the tests.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the
reference's own network code imported in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
"""
from . import warp, generator, discriminator, losses, tps  # noqa: F401
