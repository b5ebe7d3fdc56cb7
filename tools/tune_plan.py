#!/usr/bin/env python3
"""Check dat_conv3d_fwd's launch-plan model (positions per block 128 | 256, split-K factor) against measurements: every
distinct conv layer shape of the bench network is timed under the built-in plan and under every forced (bp, ksplit)
combination (dat_conv3d_tune_plan).  Prints one line per layer: the model's time, the best forced combination, the gap."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from detectandtrack_amd.ops import hip_ops as ops  # noqa
from detectandtrack_amd import libdat as L  # noqa
from bench_layers import layer_list  # noqa


def timed(layer, x, y, T, iters):
    for _ in range(2):
        layer(x, T=T, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        layer(x, T=T, out=y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='R18')
    ap.add_argument('--iters', type=int, default=8)
    ap.add_argument('--clips', type=int, default=1, help='clips per forward (bench.py --batch): frames / rois of every layer x this')
    ap.add_argument('--only', default='', help='substring filter on the layer names')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    T, H, W = 8, 768, 1344
    layers = [l for l in layer_list(a.arch, T, H, W, 3) if l[0] != 'stem_k4x1']
    # heads: RPN 3x3 conv per level on the key frame, keypoint head convs on 100 rois x 14 x 14
    for lvl in range(2, 7):
        layers.append(('conv_rpn_P%d' % lvl, 256, 256, (1, 3, 3), 1, H >> lvl, W >> lvl, 1, 1))
    layers.append(('conv_fcn', 512, 512, (1, 3, 3), 1, 14, 14, 7, 100))
    total_model = total_best = 0.0
    for l in layers:
        name, cin, cout, k, st, hi, wi, cnt = l[:8]
        if a.only and a.only not in name:
            continue
        frames = (l[8] if len(l) > 8 else T) * a.clips
        Tl = T if len(l) <= 8 else 1
        w = torch.randn(cout, cin, *k, device=dev) * (2.0 / (cin * k[0] * k[1] * k[2])) ** 0.5
        layer = ops.ConvLayer(w, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), stride=(st, st),
                              pads=(k[0] // 2, k[1] // 2, k[2] // 2), relu=True, dtype=ops.BF16)
        x = torch.randn(frames, hi, wi, layer.cin, device=dev).to(torch.bfloat16)
        y = layer(x, T=Tl)
        if x.numel() * 2 > (4 << 30):   # (the P2 post-hoc conv at 4 clips: 1 GB in + 1 GB out is fine; guard against anything larger)
            continue
        ops.tune_plan(0, 0)
        t_model = timed(layer, x, y, Tl, a.iters)
        res = {}
        for bp in (128, 256):
            for ks in (1, 2, 3, 4, 6, 8):
                ops.tune_plan(bp, ks)
                res[(bp, ks)] = timed(layer, x, y, Tl, a.iters)
        ops.tune_plan(0, 0)
        t_model = min(t_model, timed(layer, x, y, Tl, a.iters))
        (bbp, bks), t_best = min(res.items(), key=lambda kv: kv[1])
        total_model += t_model * cnt
        total_best += min(t_best, t_model) * cnt
        print('%-16s %4d->%-4d k%s s%d %4dx%-4d x%d: model %7.3f ms | best bp %3d ks %d %7.3f ms (%+5.1f %%) | %s' % (
            name, cin, cout, ''.join(map(str, k)), st, hi, wi, cnt, t_model, bbp, bks, t_best, 100 * (t_best / t_model - 1),
            ' '.join('%d/%d:%.3f' % (b, s, v) for (b, s), v in sorted(res.items()) if v < 1.15 * t_best)), flush=True)
        del x, y, layer, w
    print('TOTAL: model %.3f ms, best-per-layer %.3f ms (%.1f %%)' % (total_model, total_best, 100 * (total_best / total_model - 1)))


if __name__ == '__main__':
    main()
