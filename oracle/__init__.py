"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / torch-CPU fp32) of the reference
algorithms on the hot path named by BASELINE.json (3D Mask R-CNN keypoint
detector forward: body, FPN, RPN + proposals/NMS, RoIAlign, heads) plus the
host tracker.  Each function cites the reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import anything from here, and only as the *checker*.  The product package
`detectandtrack_amd` never imports `oracle` (a test enforces that).

PARITY STATUS (what pins each restatement; generators and fixtures are committed under tests/golden/)
  * NMS / IoU (`oracle.nms`, `oracle.boxes.bbox_overlaps`): PINNED against the reference's own Cython sources compiled from
    /root/reference into `oracle/_ref/` (`oracle/build_ref.py`) and against golden fixtures generated from that build.
  * anchors, bbox / tube transforms, GenerateProposals, RoIToBatchFormat, FPN level mapping, collect / distribute, weight
    inflation, lr policy, detection post-processing (box_results_with_nms_and_limit, soft-NMS, box voting), training labels
    (RPN labels, proposal merge, roi sampling, bbox / keypoint targets -- boxes and T = 3 tubes), the `data` blob layout and
    scale rule, the PoseTrack annorect structure and the host tracker: PINNED by golden vectors made by RUNNING the
    reference's Python (lib/...) under py3 shims -- `tests/golden/make_golden.py`.
  * the model graph (which ops, wired how, with which hyper-parameters): PINNED -- the reference's own builder functions
    (lib/modeling/*.py) executed on a recorder; `oracle.net3d` follows the same wiring.
  * AffineChannelNd: PINNED -- the reference's own CUDA operator (lib/ops/affine_channel_nd_op.cu) compiled for gfx950
    against a stand-in of the Caffe2 API it uses (`oracle/ref_affine`), run on the GPU.
  * heatmap decode (`oracle.resize.heatmaps_to_keypoints`): the function body is PINNED to lib/utils/keypoints.py:94-149 run
    with this package's INTER_CUBIC restatement in place of the absent cv2.resize; the OpenCV resamplers themselves
    (`oracle.resize`) are restated from the published algorithm and pinned by exact-rational known answers.
  * legacy RoIAlign, ConvTranspose k4 s2 p1: known answers worked by hand (tests/test_oracle_golden.py).
  * round 6: the keypoint OUTPUT function of a 3D head in both settings of KRCNN.NO_3D_DECONV_TIME_TO_CH (`Net.kps_outputs_tube`): the
    WIRING is pinned to the reference's own add_heatmap_outputs run on the recorder (tests/golden/reference_heatmap_outputs.json); what
    `group = T` computes inside Caffe2's ConvTranspose is not (both readings are restated: grouped filter / full brew filter).  The 2D
    C4 functions (`rpn_c4_2d`, `box_head_c4_2d`, `kps_head_c4_2d`: the graph of configs/video/3d/01_R-18_*.yaml) follow
    model_builder.py:500-609 (nd=False), ResNet.py:268-287 and keypoint_rcnn_heads.py:39-69.  The DATASET layer is product code pinned
    directly to the reference (tests/test_dataset_cpu.py), not an oracle restatement.
  * conv / RoIAlign / deconv / FC / loss graph as a whole (`oracle.net3d`, `oracle.train_ref`): PARITY UNPINNED -- the
    arithmetic lives in Caffe2 @ b4e1588 (+ cuDNN 7.1.2), which is not in /root/reference and cannot be built here
    (SURVEY.md F1/F7, section 8c).  The restatement follows the public operator semantics at the reference's call sites;
    there are no golden vectors for it anywhere in the reference.
"""
