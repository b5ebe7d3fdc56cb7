#!/usr/bin/env python3
"""Does a conv run slower into a freshly allocated output than into a reused one (TLB / page state)?  args like bench_one.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectandtrack_amd.ops import hip_ops as ops  # noqa

cin, cout, kt, kh, kw, st, T, H, W, iters = [int(v) for v in sys.argv[1:11]]
dev = torch.device('cuda:0')
w = torch.randn(cout, cin, kt, kh, kw, device=dev) * (2.0 / (cin * kt * kh * kw)) ** 0.5
layer = ops.ConvLayer(w, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), stride=(st, st),
                      pads=(kt // 2, kh // 2, kw // 2), relu=True, dtype=ops.BF16)
x = torch.randn(T, H, W, layer.cin, device=dev).to(torch.bfloat16)
y = layer(x, T=T)
junk = [torch.empty(64 << 20, dtype=torch.uint8, device=dev) for _ in range(64)]   # 4 GB of other live allocations


def run(mode):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    keep = []
    e0.record()
    for i in range(iters):
        if mode == 'reuse':
            layer(x, T=T, out=y)
        elif mode == 'fresh':
            keep.append(layer(x, T=T))          # a new buffer every call (kept alive: never the same block twice)
        else:
            layer(x, T=T)                       # allocator hands the same freed block back
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for mode in ('reuse', 'recycle', 'fresh', 'reuse'):
    print('%-8s %.3f ms' % (mode, run(mode)))
