#!/usr/bin/env python3
"""Sliding-window inference throughput (one clip per key frame, stride 1 — how the reference scores a video,
lib/utils/video.py:149-201) with and without the per-frame trunk cache (cfg.HIP.FRAME_TRUNK_CACHE): with the cache only the
ONE new frame of every clip runs conv1 / pool1 / res2 (and would be uploaded); results are identical
(tests/test_gpu_model.py::test_frame_trunk_cache_gives_identical_sliding_window_results).  Frames are resident in HBM,
clips strictly sequential (bench.py --pipeline 1 order).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa
from detectandtrack_amd.core.config import cfg  # noqa


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='18')
    ap.add_argument('--frames', type=int, default=8)
    ap.add_argument('--height', type=int, default=768)
    ap.add_argument('--width', type=int, default=1344)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    model, ws = bench.build(a.arch, a.frames, 'bf16', False)
    T, H, W = a.frames, a.height, a.width
    clip = bench.synthetic_clip(T, H, W, 0).cuda()
    one = clip[:, :, :1].contiguous()
    im_info = np.array([[H, W, 800.0 / 720.0]], dtype=np.float32)
    im_shape = (int(round(H / im_info[0, 2])), int(round(W / im_info[0, 2])), 3)
    out = {}
    for mode in ('plain', 'cached'):
        cfg.HIP.FRAME_TRUNK_CACHE = 2 * T if mode == 'cached' else 0
        ws.trunk_cache.clear()

        def step(s):
            if mode == 'cached':
                ids = list(range(s, s + T))
                new = ws.trunk_missing(ids)
                if new:
                    ws.FeedBlob('data', clip if len(new) == T else one)
                ws.FeedBlob('im_info', im_info)
                ws.trunk_request = (ids, new)
                ws.RunNet(model.net.name)
            else:
                bench.stage_net(model, ws, clip, im_info)
            bench.stage_heads(model, ws, im_info, im_shape)
        for s in range(a.warmup):
            step(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(a.warmup, a.warmup + a.steps):
            step(s)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out[mode] = {'clips_per_s': a.steps / el, 'ms_per_clip': 1e3 * el / a.steps}
    cfg.HIP.FRAME_TRUNK_CACHE = 0
    out['speedup'] = out['cached']['clips_per_s'] / out['plain']['clips_per_s']
    out['config'] = {'workload': '3D R-%s FPN3D keypoint R-CNN, sliding window stride 1, %dx%dx%d clips, bf16, sequential clips' % (a.arch, T, H, W)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
