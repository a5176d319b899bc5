"""CPU restatement of the reference's algorithms for the hot path — TEST INFRASTRUCTURE ONLY.

Nothing under holocron_amd/ may import this package.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg use it, as the checker (never as the thing measured or shipped).

The reference (frgfm/Holocron, pylocron 0.2.2.dev0) is pure Python over torch ops, so the
restatement is plain torch-CPU fp32 in functional style (no nn.Module reuse), each function citing
the reference file:line it follows.  It is pinned against golden vectors produced by importing the
reference itself in the authoring container (tests/golden/make_golden.py, fixtures committed under
tests/golden/), see tests/test_oracle_golden.py.

Third-party arithmetic on the path that is NOT in the reference tree: ``torchvision.ops``
(``box_area``, ``box_iou``, ``nms``; pinned only as ``torchvision>=0.15.0,<1.0.0`` in the
reference's pyproject.toml:35, not installed here).  oracle/tv_ops.py restates their published
algorithms; for ``nms`` the reference's own tests only pin trivial cases
(tests/test_models_detection.py:158-163,229-233), so NMS parity is anchored on those cases plus the
restated algorithm ("parity unpinned" beyond them).
"""
