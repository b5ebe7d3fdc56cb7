"""Times conv1 (dat_stem_conv), pool1 (dat_maxpool_hw) and the fused dat_stem_conv_pool on one clip.  GPU only."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectandtrack_amd.ops import hip_ops as ops  # noqa: E402


def timed(fn, iters):
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
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=8)
    ap.add_argument('--height', type=int, default=768)
    ap.add_argument('--width', type=int, default=1344)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(1)
    data = (torch.rand((1, 3, a.frames, a.height, a.width), generator=g) * 255 - 110).cuda()
    w = (torch.randn((64, 3, 1, 7, 7), generator=g) * 0.025).cuda()
    scale, bias = (torch.rand(64, generator=g) + 0.5).cuda(), torch.randn(64, generator=g).cuda()
    for name, dt in (('bf16', ops.BF16), ('f32', ops.F32)):
        layer = ops.StemConv(w, scale, bias, dt, relu=True)
        y = layer(data)
        es = y.element_size()
        t_conv = timed(lambda: layer(data), a.iters)
        t_pool = timed(lambda: ops.maxpool_hw(y, dt, 3, 2, 1), a.iters)
        t_fused = timed(lambda: layer.pooled(data), a.iters)
        pooled = layer.pooled(data)
        mb_in, mb_c1, mb_p1 = data.numel() * 4 / 1e6, y.numel() * es / 1e6, pooled.numel() * es / 1e6
        # the host-frame path (round 6): 4 clips of uint8 720 x 1280 frames -> pool1, blob path (dat_preprocess_frames + fused stem) vs the
        # fused stem reading the frames itself (dat_stem_conv_pool_u8)
        from detectandtrack_amd.utils import blob as blob_utils
        fr = torch.randint(0, 256, (4 * a.frames, 720, 1280, 3), dtype=torch.uint8, generator=g).cuda()
        sc = min(800. / 720, 1333. / 1280)
        means = [102.9801, 115.9465, 122.7717]
        d4, (oh, ow) = ops.preprocess_frames(fr, a.frames, sc, means, 32)
        fb = blob_utils.FrameBlob(fr, a.frames, sc, (oh, ow), tuple(d4.shape[-2:]), True, means)
        assert torch.equal(layer.pooled_u8(fb), layer.pooled(d4))
        t_pre = timed(lambda: ops.preprocess_frames(fr, a.frames, sc, means, 32, out=d4), a.iters)
        t_f4 = timed(lambda: layer.pooled(d4), a.iters)
        t_u8 = timed(lambda: layer.pooled_u8(fb), a.iters)
        print('%s: 4 clips from uint8 frames: preprocess %.1f us + fused stem %.1f us = %.1f us;  stem from the frames %.1f us (%.0f MB in instead of %.0f)'
              % (name, t_pre, t_f4, t_pre + t_f4, t_u8, fr.numel() / 1e6, d4.numel() * 4 / 1e6))
        print('%s: conv1 %.1f us (%.0f GB/s)  pool1 %.1f us (%.0f GB/s)  fused %.1f us (%.0f GB/s algorithmic: %.0f MB in, %.0f MB out)'
              % (name, t_conv, (mb_in + mb_c1) / t_conv * 1e3, t_pool, (mb_c1 + mb_p1) / t_pool * 1e3, t_fused,
                 (mb_in + mb_p1) / t_fused * 1e3, mb_in, mb_p1))


if __name__ == '__main__':
    main()
