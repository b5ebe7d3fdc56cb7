#!/usr/bin/env python3
"""Training entry point — mirror of reference tools/train_net.py:60-170.

    python tools/train_net.py --cfg configs/x.yaml [--iters N] [KEY VAL ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_net.py --cfg ...

One process per GPU; every rank trains on its own clips (NUM_GPUS in the config must equal the world size: the losses
are divided by it, model_builder.py:932-942) and the gradients are summed with bucketed RCCL all-reduces before the
identical local momentum-SGD update (training.Trainer).  The PoseTrack loader (lib/datasets, pycocotools) is not
available offline: clips and ground truth come from roi_data.synthetic (seeded boxes + 17 keypoints per person), which
has the roidb record layout, so a real loader only has to yield (frames, entry) pairs.
"""
import argparse
import glob
import logging
import os
import re
import time

import numpy as np
import torch

import _path  # noqa
from detectandtrack_amd.core.config import cfg, cfg_from_file, cfg_from_list, assert_and_infer_cfg, get_output_dir
from detectandtrack_amd.modeling import model_builder
from detectandtrack_amd.roi_data import rpn as rpn_data, fast_rcnn as frcn_data, synthetic
from detectandtrack_amd.roi_data.loader import RoIDataLoader
from detectandtrack_amd.utils import lr_policy, net as net_utils, dist as dist_utils
from detectandtrack_amd import workspace
from detectandtrack_amd.training import Trainer

logger = logging.getLogger('train_net')


def synthetic_clip_and_entry(T, h, w, seed, tube_T=1):
    rs = np.random.RandomState(seed)
    frames = rs.randint(0, 255, (1, 3, T, h // 8 + 1, w // 8 + 1)).astype(np.float32)
    data = np.repeat(np.repeat(frames, 8, axis=3), 8, axis=4)[:, :, :, :h, :w]
    data = data - np.asarray(cfg.PIXEL_MEANS, dtype=np.float32).reshape(1, 3, 1, 1, 1)
    return np.ascontiguousarray(data), synthetic.synthetic_roidb_entry(h, w, n_persons=4, seed=seed, T=tube_T)


def feed_clip(ws, data, entry, rng):
    blobs = rpn_data.add_rpn_blobs({}, 1.0, entry, rng)
    ws.FeedBlob('data', data)
    for k, v in blobs.items():
        ws.FeedBlob(k, v)
    from detectandtrack_amd.roi_data.device_sampler import make_sampler
    ws.train_sampler = make_sampler(entry, rng, seed=lambda: int(rng.randint(0, 2 ** 31 - 1)))


def main():
    logging.basicConfig(level=logging.INFO)
    p = argparse.ArgumentParser(description='Train a detection network on the MI355X')
    p.add_argument('--cfg', dest='cfg_file', required=True)
    p.add_argument('--iters', type=int, default=0, help='override SOLVER.MAX_ITER')
    p.add_argument('--height', type=int, default=256)
    p.add_argument('--width', type=int, default=320)
    p.add_argument('--roidb', default='', help='pickled roidb whose entries carry their frames as arrays (entry["image"]); default: synthetic clips')
    p.add_argument('--loader-workers', type=int, default=4,
                   help='prefetch threads of the input pipeline (roi_data.loader); 0 = label every clip synchronously on the host')
    p.add_argument('--reference-init', action='store_true', help='initialise from the builders\' init specs instead of synthetic_params')
    p.add_argument('--auto-resume', action='store_true', help='continue from the latest model_iter*.pkl of the output directory (parameters, momentum, LR schedule position)')
    p.add_argument('opts', default=None, nargs=argparse.REMAINDER)
    args = p.parse_args()
    cfg_from_file(args.cfg_file)
    if args.opts:
        cfg_from_list(args.opts)
    assert_and_infer_cfg()
    rank, _local, world = dist_utils.env_rank_world()
    torch.cuda.set_device(_local)
    dist = dist_utils.init_process_group() if world > 1 else None
    assert cfg.NUM_GPUS == world, 'NUM_GPUS (%d) must equal the number of ranks (%d)' % (cfg.NUM_GPUS, world)
    model = model_builder.create(cfg.MODEL.TYPE, train=True)
    ws = workspace.GlobalWorkspace()
    assert cfg.TRAIN.IMS_PER_BATCH == 1, \
        'one clip per forward and GPU (TRAIN.IMS_PER_BATCH 1, cf. configs/video/3d/03_*-8GPU-BATCH1.yaml:51); got %d' % cfg.TRAIN.IMS_PER_BATCH
    resume_momentum, start_iter = {}, 0
    if args.reference_init or (cfg.TRAIN.WEIGHTS and os.path.exists(cfg.TRAIN.WEIGHTS)):
        # param_init_net first (the builders' init specs, every rank alike), then overlay the file (reference train_net.py:100-118):
        # an ImageNet/COCO file has no fpn_* / rpn_* / head parameters, those keep their init
        net_utils.initialize_params(model, ws, seed=cfg.RNG_SEED)
    else:
        # no checkpoint offline: a well-conditioned deterministic init (the reference always starts from pre-trained weights;
        # its from-scratch init on raw pixels with identity AffineChannel layers diverges at the shipped learning rates)
        for k, v in net_utils.synthetic_params(model, cfg.RNG_SEED).items():
            ws.set_param(k, v)
    out_dir = get_output_dir(training=True)
    weights_file = cfg.TRAIN.WEIGHTS
    if args.auto_resume:   # latest snapshot of this output directory wins (reference train_net.py:77-98)
        snaps = sorted(glob.glob(os.path.join(out_dir, 'model_iter*.pkl')), key=lambda f: int(re.findall(r'model_iter(\d+)', f)[-1]))
        if snaps:
            weights_file = snaps[-1]
            start_iter = int(re.findall(r'model_iter(\d+)', weights_file)[-1]) + 1
            logger.info('resuming from %s at iteration %d', weights_file, start_iter)
    if weights_file and os.path.exists(weights_file):
        net_utils.initialize_from_weights_file(model, ws, weights_file, momentum=resume_momentum)
    trainer = Trainer(model, ws, dist)
    trainer.load_momentum(resume_momentum)
    T = max(cfg.VIDEO.NUM_FRAMES, 1) if cfg.MODEL.VIDEO_ON else 1
    tube_T = cfg.VIDEO.NUM_FRAMES_MID if (cfg.MODEL.VIDEO_ON and cfg.VIDEO.BODY_HEAD_LINK == '') else 1   # tube heads: 4T boxes
    rng = np.random.RandomState(cfg.RNG_SEED + rank)
    max_iter = args.iters or cfg.SOLVER.MAX_ITER
    loader = None
    # synthetic "roidb" of max_iter clips per rank (a pool of distinct pixel clips, a fresh roidb entry per index); a real
    # roidb only has to provide the same source(i) -> (data, entry, im_scale) callable
    pool = [synthetic_clip_and_entry(T, args.height, args.width, seed=1000 * rank + j, tube_T=tube_T)[0] for j in range(4)]

    def source(i):
        return pool[i % len(pool)], synthetic.synthetic_roidb_entry(args.height, args.width, n_persons=4, seed=1000 * rank + i, T=tube_T), 1.0
    n_items, sizes = max_iter, {}
    if args.roidb:
        import pickle
        from detectandtrack_amd.roi_data.minibatch import RoidbClipSource
        with open(args.roidb, 'rb') as f:
            source = RoidbClipSource(pickle.load(f), seed=cfg.RNG_SEED + rank)
        n_items, sizes = len(source), dict(widths=source.widths, heights=source.heights)
        assert args.loader_workers > 0, '--roidb needs the loader (--loader-workers > 0)'
    if args.loader_workers > 0:
        loader = RoIDataLoader(source, num_items=n_items, **sizes, num_workers=args.loader_workers, queue_size=2 * args.loader_workers,
                               device=torch.cuda.current_device(), seed=cfg.RNG_SEED + rank)
    t0 = time.time()
    it_steady = it = start_iter
    for it in range(start_iter, max_iter):
        lr = lr_policy.get_lr_at_iter(it)
        if loader is not None:
            loader.get_next_minibatch().feed(ws)
        else:
            data, entry, _ = source(it)
            feed_clip(ws, data, entry, rng)
        ex = trainer.step(lr)
        if it == min(start_iter + 5, max_iter - 1):
            torch.cuda.synchronize()
            t_steady, it_steady = time.time(), it
        if it % 20 == 0 or it == max_iter - 1:
            lv = ex.loss_values()
            logger.info('rank %d iter %d lr %.5f loss %.4f (%s) %.2f s/iter', rank, it, lr, sum(lv.values()),
                        ' '.join('%s %.3f' % (k.replace('loss_', ''), v) for k, v in sorted(lv.items())),
                        (time.time() - t0) / (it + 1))
        if rank == 0 and (it + 1) % cfg.TRAIN.SNAPSHOT_ITERS == 0:
            ws.params_from_device()
            net_utils.save_model_to_weights_file(os.path.join(out_dir, 'model_iter%d.pkl' % it), model, ws, trainer.momentum_blobs())
    torch.cuda.synchronize()
    if max_iter - 1 > it_steady:
        logger.info('rank %d steady state: %.1f ms/iter over the last %d iterations (%s)', rank,
                    1e3 * (time.time() - t_steady) / (max_iter - 1 - it_steady), max_iter - 1 - it_steady,
                    'prefetching loader, %d workers' % args.loader_workers if loader is not None else 'synchronous host labelling')
    if loader is not None:
        loader.shutdown()
    if rank == 0:
        ws.params_from_device()
        net_utils.save_model_to_weights_file(os.path.join(out_dir, 'model_final.pkl'), model, ws, trainer.momentum_blobs())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
