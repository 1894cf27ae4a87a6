"""CPU oracle for the DS-MSA / window-attention hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package imports this
directory; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may use it, and there only as the checker or the timed
CPU baseline, never as the thing that produces a shipped result.

Every function restates, op by op and in the reference's op order, what
``DSwinT_arch.py`` / ``msft_arch.py`` of jackwoodleigh/VisionTransformer do,
and cites the reference file:line it follows.  The restatement is pinned by
golden vectors produced by running the reference itself (``tests/golden``),
with one exception that is stated here and wherever it matters:

  parity unpinned: ``torchvision.ops.roi_align`` (DSwinT_arch.py:297) is a
  third-party op that is not importable in the build container, and no file in
  the reference holds outputs for it.  ``dswin_oracle.roi_align`` follows the
  algorithm torchvision publishes (RoIAlign, aligned=False, adaptive sampling
  ratio); the golden DS-MSA vectors were produced with that restatement
  standing in for torchvision.
"""
