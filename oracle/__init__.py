"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / torch-CPU fp32) of the reference
algorithms on the hot path named by BASELINE.json (3D Mask R-CNN keypoint
detector forward: body, FPN, RPN + proposals/NMS, RoIAlign, heads) plus the
host tracker.  Each function cites the reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import anything from here, and only as the *checker*.  The product package
`detectandtrack_amd` never imports `oracle` (a test enforces that).

PARITY STATUS
  * NMS / IoU (`oracle.nms`, `oracle.boxes.bbox_overlaps`): PINNED against the
    reference's own Cython sources compiled from /root/reference into
    `oracle/_ref/` (see `oracle/build_ref.py`) and against golden fixtures in
    `tests/golden/` generated from that build.
  * anchors (`oracle.anchors`): PINNED against the known-answer table in
    reference `lib/modeling/generate_anchors.py:16-39`.
  * bbox_transform: PINNED by the round-trip identity the reference tests use
    (`tests/test_bbox_transform.py:41-71`).
  * conv / affine / RoIAlign / deconv / FC graph (`oracle.net3d`): PARITY
    UNPINNED — the arithmetic lives in Caffe2 @ b4e1588 (+ cuDNN 7.1.2), which is
    not in /root/reference and cannot be built here.  The
    restatement follows the public operator semantics and the reference's
    call sites; there are no golden vectors for it anywhere in the reference.
"""
