"""smoke(): one tiny hot-path invocation on cuda:0 checked against the oracle (used by __graft_entry__)."""
import numpy as np
import torch


def run_smoke():
    from detectandtrack_amd.ops import hip_ops as ops
    from oracle.net3d import Net, opts_for
    rs = np.random.RandomState(3)
    x = rs.randn(1, 64, 2, 12, 16).astype(np.float32)
    w = (rs.randn(64, 64, 3, 3, 3) * 0.05).astype(np.float32)
    s = rs.uniform(0.5, 1.5, 64).astype(np.float32)
    b = (rs.randn(64) * 0.1).astype(np.float32)
    net = Net({'c_w': w, 'c_bn_s': s, 'c_bn_b': b}, opts_for('R18'))
    ref = torch.relu(net.conv_affine_nd(torch.from_numpy(x), 'c', [3, 3, 3], [1, 1, 1], [1, 1, 1])).numpy()
    dev = lambda a: torch.from_numpy(a).cuda()
    layer = ops.ConvLayer(dev(w), dev(s), dev(b), stride=(1, 1), pads=(1, 1, 1), relu=True, dtype=ops.F32)
    y = layer(ops.to_ndhwc(dev(x), ops.F32), T=2)
    got = ops.to_ncdhw(y, ops.F32, 1, 64, 2).cpu().numpy()
    err = float(np.abs(got - ref).max())
    assert err < 1e-3, err
    return err
