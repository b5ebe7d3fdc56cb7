"""Training parity at the BENCHED shape: one forward + backward of the FPN3D keypoint R-CNN training graph on
a full 1 x 3 x 8 x 768 x 1344 clip, R-18 and R-50 (BASELINE configs 3-4), fp32 parity mode, against torch autograd on
`oracle/train_ref.py` run on the host at the same size -- all 13 loss values and the gradient of every trainable parameter.

Reference: lib/modeling/model_builder.py:908-985 (data-parallel training graph), ResNet3D.py:21-55 (bottleneck with the stride on
the first 1x1x1, STRIDE_1X1), FPN.py:282-321 (per-level RPN losses).  At this size the weight-gradient planner picks other tiles /
split factors and the frame-window logic runs over 8 frames -- branches the 2 x 64 x 96 test of test_gpu_train.py never takes.
The sampled rois are synthetic (128 box rois, 16 keypoint rois: the oracle's per-roi Python RoIAlign is the slow part), the map
sizes -- what the conv / wgrad plans depend on -- are the benched ones.  Parity status: unpinned (oracle/train_ref.py header).
"""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FROZEN = ('conv1', 'res_conv1', 'res2_')


def _train_model(arch, T, pre, post):
    from tests.model_util import fpn3d_kps_cfg
    from detectandtrack_amd.core.config import cfg, cfg_from_cfg, assert_and_infer_cfg, reset_cfg
    from detectandtrack_amd.modeling import model_builder
    from detectandtrack_amd.utils import net as net_utils
    from detectandtrack_amd import workspace
    c = fpn3d_kps_cfg(arch, T=T, dtype='fp32', pre=pre, post=post)
    c['TRAIN'] = {'RPN_PRE_NMS_TOP_N': pre, 'RPN_POST_NMS_TOP_N': post, 'IMS_PER_BATCH': 1}
    c['NUM_GPUS'] = 1
    reset_cfg()
    cfg_from_cfg(c)
    assert_and_infer_cfg()
    model = model_builder.create(cfg.MODEL.TYPE, train=True)
    workspace.ResetWorkspace()
    ws = workspace.GlobalWorkspace()
    weights = net_utils.synthetic_params(model, 3)
    for k, v in weights.items():
        ws.set_param(k, v)
    return model, ws, weights


def run_train_step_parity(arch, T, H, W, n_rois, n_kp, pre, post, loss_rtol, worst_tol, median_tol):
    """Device forward + backward vs autograd on the oracle; returns the per-parameter relative errors."""
    from tests.model_util import synthetic_clip, oracle_opts
    from tests.test_gpu_train import _synthetic_training_blobs
    from detectandtrack_amd.core.config import cfg
    from detectandtrack_amd.training import TrainExecutor
    from oracle import train_ref
    model, ws, weights = _train_model(arch, T, pre, post)
    rs = np.random.RandomState(7)
    labels, sampled = _synthetic_training_blobs(T, H, W, rs, n_rois=n_rois, n_kp=n_kp)
    data = synthetic_clip(T, H, W)
    im_info = np.array([[H, W, 1.0]], dtype=np.float32)
    ws.FeedBlob('data', data)
    ws.FeedBlob('im_info', im_info)
    for k, v in labels.items():
        ws.FeedBlob(k, v)
    ws.train_sampler = lambda rois, info: sampled
    t0 = time.time()
    ex = TrainExecutor(ws, model.net)
    ex.run()
    ex.backward()
    got_losses = ex.loss_values()
    torch.cuda.synchronize()
    t1 = time.time()

    trainable = set(model.TrainableParams())
    # only the parameters above the StopGradient marker are leaves that need a gradient: autograd then neither records nor
    # differentiates the frozen trunk (conv1 / res2: the largest activations of the network)
    wt = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).requires_grad_(k in trainable and not k.startswith(FROZEN))
          for k, v in weights.items()}
    torch.set_num_threads(max(1, min(64, torch.get_num_threads())))
    ref_losses = train_ref.training_losses(
        wt, oracle_opts(arch, T, 3, 'slice-center', pre, post), data, im_info, labels, sampled,
        dict(num_gpus=1, rpn_batch=cfg.TRAIN.RPN_BATCH_SIZE_PER_IM, ims_per_batch=1, kps_loss_weight=cfg.KRCNN.LOSS_WEIGHT))
    sum(ref_losses.values()).backward()
    t2 = time.time()
    print('device forward+backward (fp32 parity mode, incl. first-use packing) %.1f s, oracle forward+backward %.1f s' % (t1 - t0, t2 - t1))
    assert sorted(got_losses) == sorted(ref_losses)
    for k in sorted(ref_losses):
        print('%-22s %.6f  (oracle %.6f)' % (k, got_losses[k], ref_losses[k].item()))
    for k in sorted(ref_losses):
        np.testing.assert_allclose(got_losses[k], ref_losses[k].item(), rtol=loss_rtol, atol=1e-6, err_msg=k)
    errs, bad = {}, []
    for name in sorted(trainable):
        if name.startswith(FROZEN):
            assert name not in ex.param_grads, 'gradient for a parameter below StopGradient: ' + name
            continue
        assert name in ex.param_grads, 'no gradient for ' + name
        ref = wt[name].grad
        assert ref is not None, name
        got = ex.param_grads[name].cpu()
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        assert torch.isfinite(got).all(), name
        denom = max(float(ref.abs().max()), 1e-8)
        err = float((got - ref).abs().max()) / denom
        errs[name] = err
        # the keypoint branch's gradient is sparse (a few valid keypoints per roi): an activation within 1e-7 of the ReLU threshold
        # that the two fp32 summation orders mask differently moves that layer's gradient by a few 1e-3
        tol = 6e-2 if name.startswith(('conv_fcn', 'kps_score')) else worst_tol
        if err >= tol:
            bad.append('%s: rel err %.3e (|ref|max %.3e)' % (name, err, denom))
    v = np.array(list(errs.values()))
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    print('checked gradients of %d parameters: median rel err %.2e, 90th pct %.2e, worst %s'
          % (len(v), float(np.median(v)), float(np.percentile(v, 90)), ', '.join('%s %.2e' % kv for kv in worst)))
    assert not bad, bad
    assert float(np.median(v)) < median_tol
    return errs


@pytest.mark.parametrize('arch,n_params', [('18', 40), ('50', 85)])
def test_train_step_at_the_bench_shape_matches_oracle_autograd(arch, n_params):
    """BASELINE configs 3-4 at S-C (8 x 768 x 1344): 13 losses (rtol 5e-4) + every trainable gradient (max-abs error relative to
    the gradient's max-abs < 5e-3; median over the parameters < 5e-4)."""
    errs = run_train_step_parity(arch, 8, 768, 1344, n_rois=128, n_kp=16, pre=2000, post=512,
                                 loss_rtol=5e-4, worst_tol=5e-3, median_tol=5e-4)
    assert len(errs) > n_params
    if arch == '50':     # the bottleneck layers the small-shape tests never reach: stride-2 first 1x1x1, 2048-channel layers
        for name in ('res3_0_branch2a_w', 'res3_0_branch1_w', 'res4_0_branch2a_w', 'res5_0_branch2a_w', 'res5_0_branch1_w',
                     'res5_2_branch2c_w', 'res5_2_branch2b_w', 'fpn_inner_res5_2_sum_w'):
            assert name in errs, name


# R-50's bottleneck shapes through dat_conv3d_wgrad / the data-gradient conv one layer at a time, at the benched map sizes
R50_LAYERS = [
    # name, cin, cout, (kt,kh,kw), stride, T, H, W (input map)
    ('res3_0_branch2a 1x1x1/s2 256->128', 256, 128, (1, 1, 1), 2, 8, 192, 336),
    ('res3_0_branch1 1x1x1/s2 256->512', 256, 512, (1, 1, 1), 2, 8, 192, 336),
    ('res4_0_branch2a 1x1x1/s2 512->256', 512, 256, (1, 1, 1), 2, 8, 96, 168),
    ('res5_0_branch1 1x1x1/s2 1024->2048', 1024, 2048, (1, 1, 1), 2, 8, 48, 84),
    ('res5_1_branch2a 1x1x1 2048->512', 2048, 512, (1, 1, 1), 1, 8, 24, 42),
    ('res5_1_branch2b 3x3x3 512->512', 512, 512, (3, 3, 3), 1, 8, 24, 42),
    ('res5_1_branch2c 1x1x1 512->2048', 512, 2048, (1, 1, 1), 1, 8, 24, 42),
    ('res4_1_branch2b 3x3x3 256->256', 256, 256, (3, 3, 3), 1, 8, 48, 84),
]


R50_CASES = [(l, 'fp32') for l in R50_LAYERS] + [(R50_LAYERS[i], 'bf16') for i in (1, 3, 5)]


@pytest.mark.parametrize('layer,dtype_name', R50_CASES,
                         ids=['%s_%s_%s' % (l[0].split()[0], l[0].split()[1].replace('/', ''), d) for l, d in R50_CASES])
def test_r50_bottleneck_wgrad_and_dgrad_at_the_bench_map_sizes(layer, dtype_name):
    """dat_conv3d_wgrad (+ the data gradient) of R-50's bottleneck convs at the 8 x 768 x 1344 clip's map sizes vs torch-CPU
    autograd (fp32).  bf16: operands rounded to bf16 on both sides, fp32 accumulation."""
    from detectandtrack_amd.ops import hip_ops as ops
    name, cin, cout, k, stride, T, H, W = layer
    dt = ops.F32 if dtype_name == 'fp32' else ops.BF16
    tdt = ops.tdtype(dt)
    g = torch.Generator().manual_seed(cin * 7 + cout)
    pads = (k[0] // 2, k[1] // 2, k[2] // 2)
    x5 = torch.randn((1, cin, T, H, W), generator=g) * 0.5
    w5 = torch.randn((cout, cin) + k, generator=g) * (1.0 / np.sqrt(cin * k[0] * k[1] * k[2]))
    scale = torch.rand(cout, generator=g) + 0.5
    Ho, Wo = (H + 2 * pads[1] - k[1]) // stride + 1, (W + 2 * pads[2] - k[2]) // stride + 1
    dy5 = torch.randn((1, cout, T, Ho, Wo), generator=g)
    if dtype_name == 'bf16':
        x5, dy5 = x5.bfloat16().float(), dy5.bfloat16().float()
    cs_in, cs_out = ops.round_up(cin, 64), ops.round_up(cout, 64)

    def ndhwc(v, cs):
        n, c, t, h, w = v.shape
        out = torch.zeros((n * t, h, w, cs), dtype=torch.float32)
        out[..., :c] = v.permute(0, 2, 3, 4, 1).reshape(n * t, h, w, c)
        return out.to(tdt).cuda()
    xd, gd = ndhwc(x5, cs_in), ndhwc(dy5, cs_out)
    cg = ops.ConvGrad(w5.cuda(), scale.cuda(), (stride, stride), pads, dt, cs_in, cs_out)
    dW, _ = cg.weight(xd, gd, T)
    dx = cg.data(gd, T, H, W)
    torch.cuda.synchronize()
    torch.set_num_threads(max(1, min(64, torch.get_num_threads())))
    xr = x5.clone().requires_grad_(True)
    wr = w5.clone().requires_grad_(True)
    y = torch.nn.functional.conv3d(xr, wr, None, stride=(1, stride, stride), padding=pads) * scale.view(1, -1, 1, 1, 1)
    y.backward(dy5)
    ref_dW, ref_dx = wr.grad, xr.grad
    got_dW = dW.float().cpu().reshape(ref_dW.shape)
    got_dx = dx.float().cpu().view(1, T, H, W, cs_in)[..., :cin].permute(0, 4, 1, 2, 3)
    ew = float((got_dW - ref_dW).abs().max()) / float(ref_dW.abs().max())
    ex = float((got_dx - ref_dx).abs().max()) / float(ref_dx.abs().max())
    print('%s %s: dW rel err %.2e, dx rel err %.2e' % (name, dtype_name, ew, ex))
    assert ew < (2e-4 if dtype_name == 'fp32' else 2e-3), ew
    assert ex < (2e-4 if dtype_name == 'fp32' else 2e-2), ex     # bf16: weights AND the output are rounded to bf16
