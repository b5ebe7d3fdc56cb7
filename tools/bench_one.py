#!/usr/bin/env python3
"""Run ONE conv layer shape repeatedly (for rocprofv3 --pmc runs).  args: cin cout kt kh kw stride T H W iters"""
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
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    layer(x, T=T, out=y)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print('%.3f ms  %.1f TFLOP/s' % (ms, layer.flops(T, H, W) / ms / 1e9))
