"""Cold-cache timing of 1x1 convs (64 -> Cout) at the FPN P2 resolution: how the cost of the output stores depends on the row
width a block writes.  Developer tool (GPU only).  DAT_CONV_ABLATE=4 skips the stores."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectandtrack_amd.ops import hip_ops as ops  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    T, H, W = 8, 192, 336
    junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    for cin, cout in ((64, 64), (64, 128), (64, 256), (64, 512), (128, 256), (256, 256)):
        w = torch.randn(cout, cin, 1, 1, 1, device=dev) * 0.1
        layer = ops.ConvLayer(w, None, torch.zeros(cout, device=dev), stride=(1, 1), pads=(0, 0, 0), relu=False, dtype=ops.BF16)
        x = torch.randn(T, H, W, layer.cin, device=dev).to(torch.bfloat16)
        y = layer(x, T=T)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ms = 0.0
        for it in range(8):
            junk.fill_(float(it))
            e0.record()
            layer(x, T=T, out=y)
            e1.record()
            torch.cuda.synchronize()
            ms += e0.elapsed_time(e1) / 8
        mb_in, mb_out = x.numel() * 2 / 1e6, y.numel() * 2 / 1e6
        print('%3d -> %3d: %.3f ms   in %.0f MB  out %.0f MB   %.2f TB/s' % (cin, cout, ms, mb_in, mb_out, (mb_in + mb_out) / ms / 1e9 * 1e3))


if __name__ == '__main__':
    main()
