#!/usr/bin/env python3
"""Inference entry point — mirror of reference tools/test_net.py:66-143.

    python tools/test_net.py --cfg configs/x.yaml [--range s e] [--multi-gpu-testing] [--roidb clips.pkl] [KEY VAL ...]

The clip list is cfg.TEST.DATASET read through datasets/json_dataset.py (COCO-format lists, no pycocotools; a dataset mounted elsewhere:
DAT_DATASET_ROOT / DAT_DATASET_CATALOG), like the reference; offline it can also come from `--roidb` (a pickled list of
{'image': [T frame arrays or paths], 'height', 'width'}) or `--synthetic N` / `--synthetic-video N`.  `--multi-gpu-testing` shards clips over the ranks of a torch.distributed launch
(`python -m torch.distributed.run --nproc-per-node N tools/test_net.py ...`), the reference's one-process-per-GPU
`--range` protocol (lib/utils/subprocess.py:38-63) also works unchanged.
"""
import argparse
import logging
import pickle

import numpy as np

import _path  # noqa
from detectandtrack_amd.core.config import cfg, cfg_from_file, cfg_from_list, assert_and_infer_cfg, get_output_dir
from detectandtrack_amd.core import test_engine
from detectandtrack_amd.utils.video import clip_frame_ids


def parse_args():
    p = argparse.ArgumentParser(description='Test a detection network on the MI355X')
    p.add_argument('--cfg', dest='cfg_file', required=True)
    p.add_argument('--range', dest='range', type=int, nargs=2, default=None)
    p.add_argument('--multi-gpu-testing', dest='multi_gpu_testing', action='store_true')
    p.add_argument('--roidb', default='', help='pickled clip list')
    p.add_argument('--synthetic', type=int, default=0, help='use N synthetic clips')
    p.add_argument('--synthetic-video', type=int, default=0,
                   help='one synthetic video of N frames scored like the reference scores a video: a clip around every frame, stride 1, '
                        'border frames replicated (entries carry frame_ids, so HIP.FRAME_TRUNK_CACHE can reuse the per-frame trunk)')
    p.add_argument('--synthetic-weights', action='store_true',
                   help='with no TEST.WEIGHTS: well-conditioned random weights (utils.net.synthetic_params, what bench.py uses) instead of the '
                        'builder\'s own init -- the reference init (std-0.01 score layers) gives every roi the same score, i.e. exact ties at '
                        'the detection limit for every clip, which is not what a trained model does')
    p.add_argument('opts', default=None, nargs=argparse.REMAINDER)
    return p.parse_args()


def synthetic_roidb(n, T, h=720, w=1280, seed=3):
    rs = np.random.RandomState(seed)
    return [{'image': [rs.randint(0, 255, (h, w, 3)).astype(np.uint8) for _ in range(T)], 'height': h, 'width': w,
             'name': 'images/vid%04d/%06d.jpg' % (i // 100, i % 100)} for i in range(n)]


def synthetic_video_roidb(n_frames, T, h=720, w=1280, seed=3):
    """Sliding-window clips over one video (reference lib/utils/video.py:149-201): the clip of key frame k holds frames
    k - T//2 ... k - T//2 + T - 1, clamped to the video."""
    rs = np.random.RandomState(seed)
    video = [rs.randint(0, 255, (h, w, 3)).astype(np.uint8) for _ in range(n_frames)]
    roidb = []
    for k in range(n_frames):
        ids = clip_frame_ids(k, 0, n_frames - 1, T)           # (lib/utils/video.py:149-201: one clip per key frame, borders replicated)
        roidb.append({'image': [video[i] for i in ids], 'frame_ids': [('vid0000', i) for i in ids], 'height': h, 'width': w,
                      'name': 'images/vid0000/%06d.jpg' % k})
    return roidb


def main():
    logging.basicConfig(level=logging.INFO)
    args = parse_args()
    cfg_from_file(args.cfg_file)
    if args.opts:
        cfg_from_list(args.opts)
    assert_and_infer_cfg()
    if args.roidb:
        with open(args.roidb, 'rb') as f:
            roidb = pickle.load(f)
    elif args.synthetic_video:
        roidb = synthetic_video_roidb(args.synthetic_video, max(cfg.VIDEO.NUM_FRAMES, 1))
    elif args.synthetic:
        roidb = synthetic_roidb(max(args.synthetic, 1), max(cfg.VIDEO.NUM_FRAMES, 1))
    else:       # the reference's own path (tools/test_net.py:66-143 -> core/test_engine.py:77-103): cfg.TEST.DATASET through the dataset layer
        from detectandtrack_amd.datasets.json_dataset import load_catalog_from_env
        load_catalog_from_env()
        roidb = test_engine.get_roidb_and_dataset(None)[0]
    if args.synthetic_weights:
        test_engine.SYNTHETIC_WEIGHTS = True
    out = get_output_dir(training=False)
    if args.range is not None:
        test_engine.test_net(roidb, tuple(args.range), out)
    else:
        test_engine.test_net_on_dataset(roidb, multi_gpu=args.multi_gpu_testing, output_dir=out)
    stats = getattr(test_engine.test_net, 'last_stats', None)
    if stats:       # the pipelined engine's own account of the run (steady state: the first forwards include graph capture / packing)
        import json
        print(json.dumps({'test_net': stats}))


if __name__ == '__main__':
    main()
