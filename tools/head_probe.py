"""Timing of the one-round 3x3 layers (keypoint-head conv, res4, res5, P4 shapes of the R-18 bench clip) under the environment's launch
plan -- the harness for experiments with blocks per CU / tile size / split-K (DAT_CONV_BP, DAT_CONV_KSPLIT, DAT_CONV_LDS_PAD ...).
Developer tool (GPU only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectandtrack_amd.ops import hip_ops as ops  # noqa: E402

LAYERS = [
    # name, frames(T), H, W, Cin, Cout, kt
    ('conv_fcn', int(os.environ.get('HEAD_ROIS', '100')), 14, 14, 512, 512, 1),
    ('res4_3x3x3', 8, 48, 84, 256, 256, 3),
    ('res5_3x3x3', 8, 24, 42, 512, 512, 3),
    ('fpn_post_P4', 8, 48, 84, 256, 256, 3),
    ('res3_3x3x3', 8, 96, 168, 128, 128, 3),
]


def main():
    dev = torch.device('cuda:0')
    only = sys.argv[1].split(',') if len(sys.argv) > 1 else None
    out = []
    for name, T, H, W, cin, cout, kt in LAYERS:
        if only and name not in only:
            continue
        w = torch.randn(cout, cin, kt, 3, 3, device=dev) * (2.0 / (cin * 9 * kt)) ** 0.5
        layer = ops.ConvLayer(w, None, torch.zeros(cout, device=dev), stride=(1, 1), pads=(kt // 2, 1, 1), relu=True, dtype=ops.BF16)
        x = torch.randn(T, H, W, layer.cin, device=dev).to(torch.bfloat16)
        tt = T if kt > 1 else 1
        y = layer(x, T=tt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            layer(x, T=tt, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        fl = 2.0 * cout * cin * kt * 9 * T * H * W
        out.append('%s %.3f ms %.0f TF' % (name, ms, fl / ms / 1e9))
    print('; '.join(out))


if __name__ == '__main__':
    main()
