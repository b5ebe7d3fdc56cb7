"""Times conv1x1_ks_kernel on R-50's K >= 512 pointwise layer shapes (4 clips of 8 x 768 x 1344) next to the generic kernel (forced plan) and
the DAT_CONV_ABLATE variants (1: no row copies, 2: no weight copies, 4: no MFMAs; the context reads the switch when it is created)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detectandtrack_amd.ops import hip_ops as ops  # noqa: E402

SHAPES = [('res4_2a', 1024, 256, 32, 48, 84, 1), ('res5_2a', 2048, 512, 32, 24, 42, 1), ('res5_sc_s2', 1024, 2048, 32, 48, 84, 2),
          ('res5_2c', 512, 2048, 32, 24, 42, 1), ('P4_lat', 1024, 256, 32, 48, 84, 1), ('res4_sc_s2', 512, 1024, 32, 96, 168, 2)]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    g = torch.Generator().manual_seed(1)
    for name, cin, cout, frames, h, w, stride in SHAPES:
        wt = (torch.randn((cout, cin, 1, 1, 1), generator=g) * 0.03).cuda()
        x = torch.randn((frames, h, w, cin), generator=g).to(ops.H16_DTYPE).cuda()
        layer = ops.ConvLayer(wt, torch.ones(cout).cuda(), torch.zeros(cout).cuda(), stride=(stride, stride), pads=(0, 0, 0), relu=True, dtype=ops.BF16)
        t = timed(lambda: layer(x, T=8))
        ops.tune_plan(128, 1)
        tg = timed(lambda: layer(x, T=8))
        ops.tune_plan(0, 0)
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        npos = frames * ho * wo
        mb = (npos * cin + npos * cout + cin * cout) * 2 / 1e6
        fl = 2.0 * cin * cout * npos
        print('%-12s K %4d -> %4d, %6d positions: %7.1f us  %5.2f TB/s  %6.0f TFLOP/s   (generic kernel %7.1f us)   ablate=%s'
              % (name, cin, cout, npos, t, mb / t, fl / t / 1e6, tg, os.environ.get('DAT_CONV_ABLATE', '0')))


if __name__ == '__main__':
    main()
