#!/usr/bin/env python3
"""bench.py — clips/sec of the 3D Mask R-CNN keypoint detector hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one synthetic clip that is ALREADY RESIDENT IN HBM:
`model.net` (ResNet3D body + FPN3D + FPN RPN + on-device proposals/NMS + RoIAlign + 2-MLP box head), the
reference's host glue between the nets (core/test.py:750-806: score threshold, per-class NMS on the device,
top-100) and `model.keypoint_net` (RoIAlign + 8 convs + deconv + bilinear up) up to the `kps_score` blob.
Image decoding/resizing and the host heatmap->keypoint decoding ("next") are outside the step.

Multi-GPU: clips are independent units — every rank processes its own clip stream, no
data-path collective; the only collectives are the barrier + MAX-reduce of the timing. scaling = weak.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


def model_cfg(arch, T, dtype, keyframe_dce=False):
    return {
        'MODEL': {'TYPE': 'keypoint_rcnn', 'CONV_BODY': 'FPN3D.add_fpn_ResNet%s_conv5_body' % arch,
                  'ROI_HEAD': 'head_builder.add_roi_2mlp_head', 'NUM_CLASSES': 2, 'FASTER_RCNN': True,
                  'KEYPOINTS_ON': True, 'VIDEO_ON': True},
        'FPN': {'FPN_ON': True, 'MULTILEVEL_ROIS': True, 'MULTILEVEL_RPN': True},
        'FAST_RCNN': {'ROI_XFORM_METHOD': 'RoIAlign', 'ROI_XFORM_RESOLUTION': 7, 'ROI_XFORM_SAMPLING_RATIO': 2},
        'KRCNN': {'ROI_KEYPOINTS_HEAD': 'keypoint_rcnn_heads.add_roi_pose_head_v1convX', 'NUM_STACKED_CONVS': 8,
                  'NUM_KEYPOINTS': 17, 'USE_DECONV_OUTPUT': True, 'CONV_INIT': 'MSRAFill', 'CONV_HEAD_DIM': 512,
                  'UP_SCALE': 2, 'HEATMAP_SIZE': 56, 'ROI_XFORM_METHOD': 'RoIAlign', 'ROI_XFORM_RESOLUTION': 14,
                  'ROI_XFORM_SAMPLING_RATIO': 2},
        'VIDEO': {'NUM_FRAMES': T, 'TIME_KERNEL_DIM': 3, 'BODY_HEAD_LINK': 'slice-center',
                  'WEIGHTS_INFLATE_MODE': 'center-only'},
        'TEST': {'RPN_PRE_NMS_TOP_N': 1000, 'RPN_POST_NMS_TOP_N': 1000, 'COMPETITION_MODE': False, 'NMS': 0.5,
                 'SCALES': (800,), 'MAX_SIZE': 1333},
        'HIP': {'DTYPE': dtype, 'KEYFRAME_DCE': bool(keyframe_dce)},
    }


def vendor_gemm_tflops(n=8192, dtype=torch.bfloat16):
    """What the vendor's tuned dense GEMM (torch.matmul -> hipBLASLt) reaches on THIS box right now: the practical MFMA
    ceiling of the power-capped part, reported next to the nominal 2.5 PFLOP/s (SURVEY.md §8d asks for both).  Measurement
    only, never on the product path."""
    try:
        a = torch.randn(n, n, device='cuda', dtype=dtype)
        b = torch.randn(n, n, device='cuda', dtype=dtype)
        torch.matmul(a, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.matmul(a, b)
        e1.record()
        torch.cuda.synchronize()
        return round(2.0 * n ** 3 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
    except Exception:
        return None


def pmc_traffic(a, kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
    produced by tools/prof_round.sh + tools/pmc_summary.py on this exact workload); None when the run's workload or
    dominant kernel differs from the profiled one.  bench.py cannot read PMC counters itself."""
    path = os.path.join(REPO, 'profiles', 'pmc_traffic.json')
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    w = rec.get('workload', {})
    same = (w.get('arch') == a.arch and w.get('frames') == a.frames and w.get('height') == a.height and
            w.get('width') == a.width and w.get('dtype') == a.dtype and bool(w.get('keyframe_dce')) == bool(a.keyframe_dce))
    k = rec.get('kernels', {}).get(kernel_name)
    if not same or k is None:
        return None
    return k['hbm_bytes_per_launch']


def synthetic_clip(T, H, W, seed):
    """Seeded uint8-like BGR frames minus PIXEL_MEANS, NC(T)HW fp32."""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand((1, 3, T, H // 8, W // 8), generator=g) * 255.0
    data = torch.nn.functional.interpolate(base.view(1, 3 * T, H // 8, W // 8), size=(H, W), mode='nearest').view(1, 3, T, H, W)
    data = (data + (torch.rand(data.shape, generator=g) - 0.5) * 40.0).clamp_(0, 255)
    means = torch.tensor([102.9801, 115.9465, 122.7717]).view(1, 3, 1, 1, 1)
    return (data - means).contiguous()


def build(arch, T, dtype, keyframe_dce=False):
    from detectandtrack_amd.core.config import cfg, cfg_from_cfg, assert_and_infer_cfg, reset_cfg
    from detectandtrack_amd.modeling import model_builder
    from detectandtrack_amd.utils import net as net_utils
    from detectandtrack_amd import workspace
    reset_cfg()
    cfg_from_cfg(model_cfg(arch, T, dtype, keyframe_dce))
    assert_and_infer_cfg()
    model = model_builder.create(cfg.MODEL.TYPE, train=False)
    workspace.ResetWorkspace()
    ws = workspace.GlobalWorkspace()
    for k, v in net_utils.synthetic_params(model, cfg.RNG_SEED).items():   # no checkpoints offline: random init
        ws.set_param(k, v)
    ws.CreateNet(model.net)
    ws.CreateNet(model.keypoint_net)
    return model, ws


def stage_net(model, ws, data_dev, im_info):
    """Stage A of a step: feed the resident clip and enqueue `model.net` (asynchronous)."""
    ws.FeedBlob('data', data_dev)
    ws.FeedBlob('im_info', im_info)
    ws.RunNet(model.net.name)


def stage_heads(model, ws, im_info, im_shape):
    """Stage B: the reference's host glue (core/test.py:215-252, 750-806; NMS on the device) + `keypoint_net` + the
    heatmap decode (on the device): the step ends with the per-detection 4 x 17 keypoint rows on the host."""
    from detectandtrack_amd.core import test as engine
    from detectandtrack_amd import workspace as wsmod
    prev, wsmod._GLOBAL = wsmod._GLOBAL, ws          # the engine functions talk to the global workspace
    try:
        scores, boxes, _ = engine._read_bbox_outputs([np.zeros(im_shape, np.uint8)], np.array([im_info[0, 2]]))
        scores, boxes, cls_boxes = engine.box_results_with_nms_and_limit(scores, boxes)
        n_det = boxes.shape[0]
        if n_det > 0:   # keypoint net on the detections + on-device heatmap decode (core/test.py:584-627, 865-894)
            engine.keypoint_results_on_device(model, cls_boxes, boxes, np.array([im_info[0, 2]]))
    finally:
        wsmod._GLOBAL = prev
    return n_det


class ClipPipeline(object):
    """Runs steps with up to `depth` clips in flight, each on its own HIP stream + blob namespace: while the host
    decodes/NMS-filters the boxes of clip i, the device already runs the body of clip i+1.  depth=1 is the strictly
    sequential reference order (im_detect_all per clip)."""

    def __init__(self, model, ws, depth):
        self.model, self.depth = model, depth
        self.slots = [(ws if i == 0 else ws.fork(), torch.cuda.Stream()) for i in range(depth)]
        self.pending = []
        self.n_det = 0
        self.i = 0

    def submit(self, data_dev, im_info, im_shape):
        w, st = self.slots[self.i % self.depth]
        self.i += 1
        if len(self.pending) == self.depth:
            self._finish(self.pending.pop(0))
        with torch.cuda.stream(st):
            stage_net(self.model, w, data_dev, im_info)
        self.pending.append((w, st, im_info, im_shape))

    def _finish(self, item):
        w, st, im_info, im_shape = item
        with torch.cuda.stream(st):
            self.n_det = stage_heads(self.model, w, im_info, im_shape)

    def drain(self):
        while self.pending:
            self._finish(self.pending.pop(0))
        torch.cuda.synchronize()


def cpu_baseline(arch, T, seconds_budget=15.0):
    """The oracle (torch-CPU fp32 restatement of the reference graph) timed on the host cores on a bounded sample:
    the same model on a reduced 8x256x320 clip (full clips take minutes on CPU)."""
    from detectandtrack_amd.core.config import cfg
    from detectandtrack_amd.modeling import model_builder
    from detectandtrack_amd.utils import net as net_utils
    from oracle.net3d import Net, opts_for
    from oracle import proposals as op
    H, W = 256, 320
    cores = min(os.cpu_count() or 1, 32)   # torch-CPU conv3d stops scaling well before 256 threads
    torch.set_num_threads(cores)
    model = model_builder.create(cfg.MODEL.TYPE, train=False)
    weights = net_utils.synthetic_params(model, cfg.RNG_SEED)
    data = synthetic_clip(T, H, W, 3)
    im_info = np.array([[H, W, 1.0]], np.float32)
    opts = opts_for('R' + arch, kt_body=3, body_head_link='slice-center', num_frames_mid=T, pre_nms_topn=1000,
                    post_nms_topn=50)
    n, t0 = 0, time.time()
    while True:
        net = Net(weights, opts)
        net.body(data)
        p2d = net.time_link(net.fpn())
        rois, per_level, restore = net.fpn_rpn(p2d, im_info)
        feat = net.roi_feat_fpn(p2d[1:], per_level, restore, 7, 2)
        net.box_head_2mlp(feat)
        kp = rois[:4]
        _, pl, rs = op.distribute(kp, 2, 5)
        net.kps_head_2d(net.roi_feat_fpn(p2d[1:], pl, rs, 14, 2))
        n += 1
        el = time.time() - t0
        if el > seconds_budget or n >= 64:
            break
    return {'value': n / el, 'unit': 'clips/s (reduced %dx%dx%d clips)' % (T, H, W), 'cores': cores, 'kind': 'port',
            'sample': '%d forward passes of oracle.net3d (torch-CPU fp32, %d threads) on a %dx%dx%d clip, '
                      '50 rois, 4 keypoint rois; %.1f s' % (n, cores, T, H, W, el)}


def cpu_tracker_baseline():
    """Host Hungarian tracker (stays on the host by design, tools/compute_tracks.py) on the synthetic detection
    set of BASELINE.md §4: 50 videos x 100 frames x ~8 persons, single core as the reference runs it."""
    from detectandtrack_amd.core import tracking_engine as te
    return te.benchmark_synthetic(n_videos=50, n_frames=100, n_persons=8, seed=3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--arch', default='18', choices=['18', '50', '101'])
    ap.add_argument('--frames', type=int, default=8)
    ap.add_argument('--height', type=int, default=768)
    ap.add_argument('--width', type=int, default=1344)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dump-convs', action='store_true', help='per-layer conv timing to stderr')
    ap.add_argument('--pipeline', type=int, default=3, help='clips in flight per GPU (1 = strictly sequential)')
    ap.add_argument('--keyframe-dce', action='store_true',
                    help='opt-in cfg.HIP.KEYFRAME_DCE: compute only the centre frame of the FPN outputs that slice-center keeps '
                         '(identical detections; NOT the default, the default materialises every frame like the reference)')
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert world == a.gpus or world == 1, 'launch with --nproc-per-node == --gpus'
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl')

    from detectandtrack_amd.ops import hip_ops as ops
    model, ws = build(a.arch, a.frames, a.dtype, a.keyframe_dce)
    T, H, W = a.frames, a.height, a.width
    # every rank gets its own clips (weak scaling): seed by rank
    clips = [synthetic_clip(T, H, W, 1000 * rank + i).cuda() for i in range(2)]
    im_info = np.array([[H, W, 800.0 / 720.0]], dtype=np.float32)
    im_shape = (int(round(H / im_info[0, 2])), int(round(W / im_info[0, 2])), 3)

    pipe = ClipPipeline(model, ws, a.pipeline)
    torch.cuda.synchronize()
    for i in range(a.warmup):
        pipe.submit(clips[i % 2], im_info, im_shape)
    pipe.drain()

    # ---- timed region: EXACTLY `steps` steps, barrier + synchronize on both sides ----
    w0, st0 = pipe.slots[0]
    in_region = a.pipeline == 1     # per-launch events inside the timed region only when clips do not overlap
    prof = ops.ConvProfiler(capacity=256 * max(a.steps, 1))
    if in_region:
        w0.conv_log = []
        with torch.cuda.stream(st0):
            prof.start()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        pipe.submit(clips[i % 2], im_info, im_shape)
    pipe.drain()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    n_det = pipe.n_det
    prof_steps = a.steps
    if not in_region:
        # with >1 clip in flight a launch's event pair would also span the other stream's kernels: measure the
        # per-launch durations on the same clips right after the timed region, one clip at a time
        prof_steps = min(a.steps, 5)
        seq = ClipPipeline(model, w0, 1)
        seq.slots = [(w0, st0)]
        w0.conv_log = []
        with torch.cuda.stream(st0):
            prof.start()
        for i in range(prof_steps):
            seq.submit(clips[i % 2], im_info, im_shape)
        seq.drain()
    with torch.cuda.stream(st0):
        records = prof.stop()
    conv_log, w0.conv_log = (w0.conv_log or []), None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (conv3d_igemm, the BN=128 instantiation) from per-launch HIP events ----
    assert len(records) == len(conv_log), (len(records), len(conv_log))
    # dominant kernel = the conv3d_igemm instantiation with the largest total time in the timed region
    by_tag = {}
    for (tag, _, ms), (_, fl, nbytes) in zip(records, conv_log):
        t = by_tag.setdefault(tag, [0.0, 0.0, 0, 0.0])
        t[0] += fl
        t[1] += ms
        t[2] += 1
        t[3] += nbytes
    dom_tag = max(by_tag, key=lambda k: by_tag[k][1])
    dom_fl, dom_ms, dom_n, dom_bytes = by_tag[dom_tag]
    if a.dump_convs:
        agg = {}
        for (tag, _, ms), (name, fl, _b) in zip(records, conv_log):
            e = agg.setdefault(name, [0.0, 0.0, tag])
            e[0] += fl
            e[1] += ms
        for name, (fl, ms, tag) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print('%-40s tag %7d %8.3f ms/step %8.1f TFLOP/s' % (name, tag, ms / prof_steps, fl / ms / 1e9 if ms > 0 else 0),
                  file=sys.stderr)
    all_fl = sum(c[1] for c in conv_log)
    all_ms = sum(ms for _, _, ms in records)
    peak = PEAK_BF16_TFLOPS if a.dtype == 'bf16' else PEAK_F32_TFLOPS
    achieved = dom_fl / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    kernel_name = 'conv3d_igemm_kernel<%s,%d,%d>' % (a.dtype, dom_tag // 10000, (dom_tag % 10000) // 10)
    roofline = {
        'bound': 'mfma', 'kernel': kernel_name,
        'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
        'traffic': pmc_traffic(a, kernel_name),
        'algorithmic_bytes_per_launch': round(dom_bytes / max(dom_n, 1)),
        # the part runs its MFMA kernels far below the 2.4 GHz the 2.5 PFLOP/s peak assumes: clock measured inside the
        # conv kernel (s_memtime / s_memrealtime) over the profiled launches, and the dense peak rescaled to it
        'shader_clock_mhz': round(getattr(prof, 'shader_mhz', 0.0), 1),
        'peak_at_measured_clock': round(peak * getattr(prof, 'shader_mhz', 0.0) / 2400.0, 1),
        'frac_at_measured_clock': round(achieved / (peak * prof.shader_mhz / 2400.0), 4) if getattr(prof, 'shader_mhz', 0.0) > 0 else None,
        # the vendor's tuned dense bf16 GEMM on this very box (hipBLASLt 8192^3): the practical, power-capped MFMA ceiling
        'vendor_gemm_tflops_same_box': vendor_gemm_tflops() if a.dtype == 'bf16' else None,
        'launches_per_step': dom_n // max(prof_steps, 1),
        'avg_launch_ms': round(dom_ms / max(dom_n, 1), 4),
        'algorithmic_tflop_per_step': round(dom_fl / max(prof_steps, 1) / 1e12, 4),
        'all_conv_kernels': {'tflop_per_step': round(all_fl / max(prof_steps, 1) / 1e12, 4),
                             'ms_per_step': round(all_ms / max(prof_steps, 1), 3),
                             'tflops': round(all_fl / (all_ms * 1e-3) / 1e12, 2) if all_ms > 0 else 0.0},
    }
    out = {
        'metric': 'clips/sec (8-frame 800px)', 'value': round(a.gpus * a.steps / elapsed, 4), 'unit': 'clips/s',
        'n_gpus': a.gpus, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * elapsed / a.steps, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
        'config': {'workload': '3D R-%s FPN3D keypoint R-CNN inference, 1x3x%dx%dx%d clip per step per GPU '
                               '(kT=3 body+FPN, slice-center 2D heads, 1000 proposals, %d detections -> kps_score -> decoded keypoints)'
                               % (a.arch, T, H, W, n_det),
                   'weights': 'random-init (synthetic_params, seed 3)', 'clips_per_step_per_gpu': 1,
                   'clips_in_flight': a.pipeline, 'keyframe_dce': bool(a.keyframe_dce),
                   'parallelism': 'clip-sharded x%d (no data-path collective)' % a.gpus},
        'roofline': roofline,
    }
    if not a.no_cpu_baseline and a.gpus == 1:     # the CPU baseline is timed on rank 0 of the single-GPU run only
        out['cpu_baseline'] = cpu_baseline(a.arch, T)
        out['cpu_tracker'] = cpu_tracker_baseline()
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
