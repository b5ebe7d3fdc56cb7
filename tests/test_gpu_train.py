"""GPU parity of the training-side kernels against torch autograd on the CPU (fp32): weight and data
gradients of the fused conv, ReLU/bias backward, FPN top-down backward, momentum SGD."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from detectandtrack_amd.ops import hip_ops
    return hip_ops


def _ndhwc(x5, cs, dtype):
    """(N, C, T, H, W) fp32 numpy/torch -> CUDA [N*T, H, W, cs]"""
    n, c, t, h, w = x5.shape
    out = torch.zeros((n * t, h, w, cs), dtype=torch.float32)
    out[..., :c] = x5.permute(0, 2, 3, 4, 1).reshape(n * t, h, w, c)
    return out.to(dtype).cuda()


def _from_ndhwc(y, n, c, t):
    f, h, w, cs = y.shape
    return y.float().cpu().view(n, t, h, w, cs)[..., :c].permute(0, 4, 1, 2, 3).contiguous()


CASES = [
    # cin, cout, (kt,kh,kw), stride, (pt,ph,pw), N, T, H, W
    (64, 128, (3, 3, 3), 1, (1, 1, 1), 1, 3, 12, 14),
    (24, 40, (1, 3, 3), 1, (0, 1, 1), 2, 2, 9, 11),
    (64, 128, (3, 3, 3), 2, (1, 1, 1), 1, 3, 12, 16),
    (128, 256, (1, 1, 1), 2, (0, 0, 0), 1, 2, 12, 16),
    (256, 12, (1, 1, 1), 1, (0, 0, 0), 1, 2, 10, 12),
    (130, 70, (3, 3, 3), 1, (1, 1, 1), 1, 4, 8, 10),
]


@pytest.mark.parametrize('dtype_name', ['fp32', 'bf16'])
@pytest.mark.parametrize('case', CASES)
def test_conv_backward_matches_autograd(ops, case, dtype_name):
    cin, cout, k, st, pads, N, T, H, W = case
    dt = ops.F32 if dtype_name == 'fp32' else ops.BF16
    tdt = ops.tdtype(dt)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((N, cin, T, H, W), generator=g)
    w = torch.randn((cout, cin) + k, generator=g) * (2.0 / (cin * k[0] * k[1] * k[2])) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    if dtype_name == 'bf16':   # the reference sees the same rounded operands
        x = x.to(torch.bfloat16).float()
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    sr = scale.clone().requires_grad_(True)
    z = F.conv3d(xr, wr, None, stride=(1, st, st), padding=pads)
    y = z * sr.view(1, -1, 1, 1, 1)
    gy = torch.randn(y.shape, generator=g)
    if dtype_name == 'bf16':
        gy = gy.to(torch.bfloat16).float()
    y.backward(gy)

    cs_x, cs_g = ops.round_up(cin, 64), ops.round_up(cout, 64)
    xd = _ndhwc(x, cs_x, tdt)
    gd = _ndhwc(gy, cs_g, tdt)
    cg = ops.ConvGrad(w.cuda(), scale.cuda(), (st, st), pads, dt, cs_x, cs_g)
    dW, dscale = cg.weight(xd, gd, T)
    tol = 2e-4 if dtype_name == 'fp32' else 2e-2
    ref_dw = wr.grad
    err = (dW.cpu() - ref_dw).abs().max() / max(ref_dw.abs().max(), 1e-6)
    assert err < tol, 'dW rel err %.3e' % err
    err = (dscale.cpu() - sr.grad).abs().max() / max(sr.grad.abs().max(), 1e-6)
    assert err < tol, 'dscale rel err %.3e' % err
    dx = cg.data(gd, T, H, W)
    got = _from_ndhwc(dx, N, cin, T)
    err = (got - xr.grad).abs().max() / max(xr.grad.abs().max(), 1e-6)
    assert err < (2e-4 if dtype_name == 'fp32' else 3e-2), 'dx rel err %.3e' % err
    # accumulate into an existing gradient (a blob with two consumers)
    base = torch.randn(dx.shape, generator=g).to(tdt).cuda()
    base[..., cin:] = 0
    expect = base.float() + dx.float()
    acc = cg.data(gd, T, H, W, accumulate_into=base.clone())
    assert (acc.float() - expect).abs().max() <= (1e-5 if dtype_name == 'fp32' else 0.06 * expect.abs().max())


@pytest.mark.parametrize('dtype_name', ['fp32', 'bf16'])
def test_relu_bias_bwd_and_upsample_bwd(ops, dtype_name):
    dt = ops.F32 if dtype_name == 'fp32' else ops.BF16
    tdt = ops.tdtype(dt)
    g = torch.Generator().manual_seed(2)
    for cs, C in ((64, 64), (256, 200), (2048, 2048)):
        y = torch.relu(torch.randn((3, 6, 8, cs), generator=g)).to(tdt)
        dy = torch.randn((3, 6, 8, cs), generator=g).to(tdt)
        dy2 = torch.randn((3, 6, 8, cs), generator=g).to(tdt)
        dbias = torch.zeros(C, dtype=torch.float32).cuda()
        got = ops.relu_bias_bwd(dy.cuda(), y.cuda(), dt, C, relu=True, dy2=dy2.cuda(), dbias=dbias)
        ref = (dy.float() + dy2.float()) * (y.float() > 0)
        ref[..., C:] = 0
        ref_q = ref.to(tdt).float()
        assert (got.float().cpu() - ref_q).abs().max() <= (0 if dtype_name == 'fp32' else 0.02)
        np.testing.assert_allclose(dbias.cpu().numpy(), ref.reshape(-1, cs).sum(0)[:C].numpy(), rtol=1e-4, atol=1e-3)
    gfine = torch.randn((2, 8, 12, 64), generator=g).to(tdt)
    top = ops.upsample2x_bwd(gfine.cuda(), dt)
    ref = gfine.float().view(2, 4, 2, 6, 2, 64).sum(dim=(2, 4))
    assert (top.float().cpu() - ref).abs().max() <= (1e-6 if dtype_name == 'fp32' else 0.05)
    top2 = ops.upsample2x_bwd(gfine.cuda(), dt, dtop=top.clone())
    assert (top2.float().cpu() - 2 * ref).abs().max() <= (1e-5 if dtype_name == 'fp32' else 0.1)


def test_sgd_momentum_matches_reference_update(ops):
    """model_builder.py:954-985: biases grad*2 no decay; weights grad += wd*w; v = mu*v + lr*g; w -= v."""
    g = torch.Generator().manual_seed(1)
    for is_bias in (0, 1):
        w = torch.randn(1000, generator=g)
        v = torch.randn(1000, generator=g) * 0.1
        grad = torch.randn(1000, generator=g)
        lr, mu, wd = 0.01, 0.9, 1e-4
        gg = 2 * grad if is_bias else grad + wd * w
        nv = mu * v + lr * gg
        nw = w - nv
        wd_, vd, gd = w.clone().cuda(), v.clone().cuda(), grad.clone().cuda()
        ops.sgd_momentum(wd_, vd, gd, lr, mu, wd, is_bias)
        np.testing.assert_allclose(vd.cpu().numpy(), nv.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(wd_.cpu().numpy(), nw.numpy(), rtol=1e-6, atol=1e-7)
