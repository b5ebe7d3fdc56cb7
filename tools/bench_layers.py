#!/usr/bin/env python3
"""Per-layer timing of the conv3d implicit-GEMM kernel on the body+FPN layer shapes of
3D R-18 / R-50 FPN3D at clip size 1x3xTx768x1344.  Developer tool: prints one
line per distinct layer shape with ms and algorithmic TFLOP/s (no padding counted)."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectandtrack_amd.ops import hip_ops as ops  # noqa


def layer_list(arch, T, H, W, kt):
    """(name, Cin, Cout, (kt,kh,kw), stride, H_in, W_in, count)"""
    L = []
    h2, w2 = H // 4, W // 4
    L.append(('stem_k4x1', 64, 64, (1, 4, 1), 1, H // 2 + 3, W // 2, 1))
    if arch == 'R18':
        dims = (64, 64, 128, 256, 512)
        L.append(('res2_3x3', 64, 64, (1, 3, 3), 1, h2, w2, 4))
        for s in range(2, 5):
            hi, wi = h2 >> (s - 2), w2 >> (s - 2)
            cin, cout = dims[s - 1], dims[s]
            L.append(('res%d_0_2a_s2' % (s + 1), cin, cout, (kt, 3, 3), 2, hi, wi, 1))
            L.append(('res%d_sc_1x1_s2' % (s + 1), cin, cout, (1, 1, 1), 2, hi, wi, 1))
            L.append(('res%d_3x3x3' % (s + 1), cout, cout, (kt, 3, 3), 1, hi // 2, wi // 2, 3))
        lat = (512, 256, 128, 64)
    else:
        counts = (3, 4, 6, 3)
        dims = (64, 256, 512, 1024, 2048)
        inner = (64, 128, 256, 512)
        for s in range(4):
            hi, wi = h2 >> max(s - 1, 0), w2 >> max(s - 1, 0)   # input spatial of the stage
            ho, wo = h2 >> s, w2 >> s
            st = 1 if s == 0 else 2
            L.append(('res%d_0_2a' % (s + 2), dims[s], inner[s], (1, 1, 1), st, hi, wi, 1))
            L.append(('res%d_sc' % (s + 2), dims[s], dims[s + 1], (1, 1, 1), st, hi, wi, 1))
            L.append(('res%d_2a' % (s + 2), dims[s + 1], inner[s], (1, 1, 1), 1, ho, wo, counts[s] - 1))
            L.append(('res%d_2b' % (s + 2), inner[s], inner[s], (kt if s > 0 else 1, 3, 3), 1, ho, wo, counts[s]))
            L.append(('res%d_2c' % (s + 2), inner[s], dims[s + 1], (1, 1, 1), 1, ho, wo, counts[s]))
        lat = (2048, 1024, 512, 256)
    for i, c in enumerate(lat):
        hi, wi = h2 >> (3 - i), w2 >> (3 - i)
        L.append(('fpn_lat_P%d' % (5 - i), c, 256, (1, 1, 1), 1, hi, wi, 1))
        L.append(('fpn_post_P%d' % (5 - i), 256, 256, (kt, 3, 3), 1, hi, wi, 1))
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='R18')
    ap.add_argument('--T', type=int, default=8)
    ap.add_argument('--H', type=int, default=768)
    ap.add_argument('--W', type=int, default=1344)
    ap.add_argument('--kt', type=int, default=3)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--out', default='')
    ap.add_argument('--only', default='', help='comma-separated layer names')
    ap.add_argument('--cold', action='store_true', help='flush the caches (a 1-GB fill) before every timed launch: the in-network condition of HBM-bound layers')
    ap.add_argument('--topdown', action='store_true', help='fpn_lat_* layers add the nearest-2x up-sampled coarser map (res_mode 2), as in the network')
    ap.add_argument('--clock', action='store_true', help='also report the shader clock under each layer (dat_prof_clock)')
    a = ap.parse_args()
    dt = ops.BF16 if a.dtype == 'bf16' else ops.F32
    dev = torch.device('cuda:0')
    rows = []
    tot_ms = tot_fl = 0.0
    for (name, cin, cout, k, st, hi, wi, cnt) in layer_list(a.arch, a.T, a.H, a.W, a.kt):
        if a.only and name not in a.only.split(','):
            continue
        w = torch.randn(cout, cin, *k, device=dev) * (2.0 / (cin * k[0] * k[1] * k[2])) ** 0.5
        pads = (k[0] // 2, k[1] // 2, k[2] // 2) if name != 'stem_k4x1' else (0, 0, 0)
        layer = ops.ConvLayer(w, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), stride=(st, st),
                              pads=pads, relu=True, dtype=dt)
        x = torch.randn(a.T, hi, wi, layer.cin, device=dev).to(ops.tdtype(dt))
        kw = {}
        if a.topdown and name.startswith('fpn_lat_') and hi % 2 == 0 and wi % 2 == 0:
            kw = dict(residual=torch.randn(a.T, hi // 2, wi // 2, layer.cstride, device=dev).to(ops.tdtype(dt)), res_mode=2)
        y = layer(x, T=a.T, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if a.cold:
            junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)
            ms = 0.0
            for it in range(a.iters):
                junk.fill_(float(it))
                e0.record()
                layer(x, T=a.T, out=y, **kw)
                e1.record()
                torch.cuda.synchronize()
                ms += e0.elapsed_time(e1) / a.iters
            del junk
        else:
            e0.record()
            for _ in range(a.iters):
                layer(x, T=a.T, out=y, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
        fl = layer.flops(a.T, hi, wi)
        tf = fl / ms / 1e9
        mhz = 0.0
        if a.clock:
            prof = ops.ConvProfiler(64)
            prof.start()
            for _ in range(a.iters):
                layer(x, T=a.T, out=y)
            torch.cuda.synchronize()
            prof.stop()
            mhz = prof.shader_mhz
        rows.append(dict(layer=name, cin=cin, cout=cout, k=k, stride=st, hw=(hi, wi), count=cnt, ms=ms, tflops=tf))
        tot_ms += ms * cnt
        tot_fl += fl * cnt
        print('%-18s cin %4d cout %4d k %s s%d in %4dx%-4d x%d : %8.3f ms  %7.1f TFLOP/s%s' %
              (name, cin, cout, k, st, hi, wi, cnt, ms, tf, '  %.0f MHz' % mhz if a.clock else ''), flush=True)
        del x, y, layer, w
    print('TOTAL conv: %.3f ms, %.3f TFLOP -> %.1f TFLOP/s (%.1f%% of 2500)' %
          (tot_ms, tot_fl / 1e12, tot_fl / tot_ms / 1e9, tot_fl / tot_ms / 1e9 / 25.0))
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(dict(rows=rows, total_ms=tot_ms, total_tflop=tot_fl / 1e12), f, indent=1)


if __name__ == '__main__':
    main()
