"""CPU oracle for the self-supervised depth loss hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch fp32 restatement of
the reference's algorithm for the path named in SURVEY.md section 8 (geometry ->
bilinear warp -> SSIM + robust-L1 photometric loss -> per-pixel min-reprojection
-> edge-aware smoothness, plus the auxiliary TripleD loss terms).  It exists to
check the hand-written HIP kernels and to provide the timed CPU baseline.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product package never does: it fails loudly
when ``libtripled_hip.so`` is missing.

Parity status: PINNED.  Every function here is checked in
``tests/test_oracle_golden.py`` against vectors produced by importing and
running the reference's own Python modules (``tools/gen_golden.py``; the
reference has no tests or golden files of its own).

Each function cites the reference file:line it follows (paths relative to the
reference checkout).
"""

from . import geometry, photometric, smooth, metrics  # noqa: F401
