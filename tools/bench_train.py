#!/usr/bin/env python3
"""Time full training iterations (forward + losses + backward + all-reduce + SGD) on synthetic clips.
    python tools/bench_train.py --arch 18 --frames 8 --height 768 --width 1344 --iters 5
Developer tool for SURVEY.md §8d config 4 (3D R-50 FPN T=8 data-parallel training); prints ms/iter and, with
--breakdown, the kernel-time split of one iteration from torch profiler-free HIP events around forward/backward/update."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa  (model_cfg)
from detectandtrack_amd.core.config import cfg, cfg_from_cfg, assert_and_infer_cfg, reset_cfg  # noqa
from detectandtrack_amd.modeling import model_builder  # noqa
from detectandtrack_amd.utils import net as net_utils  # noqa
from detectandtrack_amd import workspace  # noqa
from detectandtrack_amd.training import Trainer, TrainExecutor  # noqa
from detectandtrack_amd.roi_data import rpn as rpn_data, fast_rcnn as frcn_data, synthetic  # noqa


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='18')
    ap.add_argument('--frames', type=int, default=8)
    ap.add_argument('--height', type=int, default=768)
    ap.add_argument('--width', type=int, default=1344)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--dtype', default='bf16')
    a = ap.parse_args()
    c = bench.model_cfg(a.arch, a.frames, a.dtype)
    c['TRAIN'] = {'RPN_PRE_NMS_TOP_N': 2000, 'RPN_POST_NMS_TOP_N': 2000, 'IMS_PER_BATCH': 1, 'MAX_SIZE': 1344,
                  'BATCH_SIZE_PER_IM': 512}
    c['NUM_GPUS'] = 1
    reset_cfg()
    cfg_from_cfg(c)
    assert_and_infer_cfg()
    model = model_builder.create(cfg.MODEL.TYPE, train=True)
    workspace.ResetWorkspace()
    ws = workspace.GlobalWorkspace()
    for k, v in net_utils.synthetic_params(model, 3).items():
        ws.set_param(k, v)
    T, H, W = a.frames, a.height, a.width
    data = bench.synthetic_clip(T, H, W, 1).cuda()
    entry = synthetic.synthetic_roidb_entry(H, W, n_persons=8, seed=1)
    rng = np.random.RandomState(0)
    blobs = rpn_data.add_rpn_blobs({}, 1.0, entry, rng)
    ws.FeedBlob('data', data)
    for k, v in blobs.items():
        ws.FeedBlob(k, v)
    from detectandtrack_amd.roi_data.device_sampler import make_sampler
    ws.train_sampler = make_sampler(entry, rng, seed=1)
    trainer = Trainer(model, ws)
    for _ in range(2):
        ex = trainer.step(1e-4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        ex = trainer.step(1e-4)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    # forward / backward split of one more iteration
    ex = TrainExecutor(ws, model.net)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ex.run()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ex.backward()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    lv = ex.loss_values()
    print('R-%s %dx%dx%d %s: %.1f ms/iter (%.2f clips/s); forward+losses %.1f ms, backward %.1f ms; loss %.3f; peak mem %.1f GB'
          % (a.arch, T, H, W, a.dtype, 1e3 * dt, 1.0 / dt, 1e3 * (t2 - t1), 1e3 * (t3 - t2), sum(lv.values()),
             torch.cuda.max_memory_allocated() / 2 ** 30))


if __name__ == '__main__':
    main()
