import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from detectandtrack_amd.ops import hip_ops as ops
def run(cin, cout, k, st, pads, N, T, H, W):
    g = torch.Generator().manual_seed(5)
    x = torch.randn((N, cin, T, H, W), generator=g).bfloat16().float()
    w = torch.randn((cout, cin) + k, generator=g) * 0.1
    Ho, Wo = (H + 2*pads[1] - k[1])//st + 1, (W + 2*pads[2] - k[2])//st + 1
    gy = torch.randn((N, cout, T, Ho, Wo), generator=g).bfloat16().float()
    xr = x.clone(); wr = w.clone().requires_grad_(True)
    y = torch.nn.functional.conv3d(xr, wr, None, stride=(1, st, st), padding=pads); y.backward(gy)
    def nd(v, cs):
        n, c, t, h, w_ = v.shape
        out = torch.zeros((n*t, h, w_, cs)); out[..., :c] = v.permute(0, 2, 3, 4, 1).reshape(n*t, h, w_, c)
        return out.bfloat16().cuda()
    csx, csg = ops.round_up(cin, 64), ops.round_up(cout, 64)
    cg = ops.ConvGrad(w.cuda(), None, (st, st), pads, ops.BF16, csx, csg)
    dW, _ = cg.weight(nd(x, csx), nd(gy, csg), T)
    dW = dW.cpu(); ref = wr.grad
    err = (dW - ref).abs()
    print(cin, cout, k, st, 'max ref', float(ref.abs().max()), 'max err', float(err.max()), 'nan', int(torch.isnan(dW).sum()), 'big', int((dW.abs() > 1e6).sum()), 'of', dW.numel())
    bad = (err > 0.05 * ref.abs().max()).nonzero()
    print('bad count', len(bad), bad[:12].tolist())
    print('sample got', dW.flatten()[:6].tolist(), 'ref', ref.flatten()[:6].tolist())
run(64, 64, (1, 1, 1), 1, (0, 0, 0), 1, 1, 8, 8)
run(64, 128, (1, 1, 1), 1, (0, 0, 0), 1, 2, 12, 14)
run(64, 128, (3, 3, 3), 1, (1, 1, 1), 1, 3, 12, 14)
run(128, 256, (1, 1, 1), 2, (0, 0, 0), 1, 2, 12, 16)
run(128, 128, (3, 3, 3), 1, (1, 1, 1), 1, 3, 13, 19)
run(64, 192, (1, 3, 3), 1, (0, 1, 1), 2, 2, 9, 11)
run(130, 70, (3, 3, 3), 1, (1, 1, 1), 1, 4, 8, 10)
