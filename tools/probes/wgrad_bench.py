"""Per-layer timing of dat_conv3d_wgrad at the training bench's layer shapes (bf16): direct kernels vs the re-pack path, K-split sweep."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from detectandtrack_amd.ops import hip_ops as ops

LAYERS = [  # name, cin, cout, k, stride, T, H, W (input), g frames window (t0, n) or None
    ('res3 3x3x3 128', 128, 128, (3, 3, 3), 1, 8, 96, 168, None),
    ('res4 3x3x3 256', 256, 256, (3, 3, 3), 1, 8, 48, 84, None),
    ('res5 3x3x3 512', 512, 512, (3, 3, 3), 1, 8, 24, 42, None),
    ('fpn P2 posthoc 1-frame g', 256, 256, (3, 3, 3), 1, 3, 192, 336, (1, 1)),
    ('fpn P3 posthoc 1-frame g', 256, 256, (3, 3, 3), 1, 3, 96, 168, (1, 1)),
    ('res4_0 2a 3x3x3 s2', 128, 256, (3, 3, 3), 2, 8, 96, 168, None),
    ('lateral P3 1x1 128->256', 128, 256, (1, 1, 1), 1, 3, 96, 168, None),
    ('r50 res3 2c 1x1 128->512', 128, 512, (1, 1, 1), 1, 8, 96, 168, None),
    ('r50 res4 2a 1x1 1024->256', 1024, 256, (1, 1, 1), 1, 8, 48, 84, None),
]

RESULTS = {}

def run(tag):
    for name, cin, cout, k, st, T, H, W, win in LAYERS:
        g = torch.Generator().manual_seed(1)
        pads = (k[0] // 2, k[1] // 2, k[2] // 2)
        Ho, Wo = (H + 2 * pads[1] - k[1]) // st + 1, (W + 2 * pads[2] - k[2]) // st + 1
        x = torch.randn((T, H, W, cin), generator=g).bfloat16().cuda()
        gy = torch.randn((T, Ho, Wo, cout), generator=g).bfloat16().cuda()
        w = torch.randn((cout, cin) + k, generator=g).cuda() * 0.05
        cg = ops.ConvGrad(w, None, (st, st), pads, ops.BF16, cin, cout)
        kw = dict(g_frames=win) if win else {}
        dW, _ = cg.weight(x, gy, T, **kw)
        torch.cuda.synchronize()
        RESULTS.setdefault(name, []).append((tag, dW.clone()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            cg.weight(x, gy, T, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        frames = win[1] if win else T
        fl = 2.0 * cin * cout * k[0] * k[1] * k[2] * frames * Ho * Wo
        print('%-8s %-28s %7.3f ms  %7.1f TFLOP/s' % (tag, name, ms, fl / ms / 1e9), flush=True)

def fresh(env, tag):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with torch.cuda.stream(torch.cuda.Stream()):
            ops.drop_ctx()          # (pooled stream handles recur: make sure the context is created under THIS environment)
            run(tag)
            torch.cuda.synchronize()
            ops.drop_ctx()
    finally:
        for k, v in old.items():
            if v is None: del os.environ[k]
            else: os.environ[k] = v

import sys as _s
if len(_s.argv) > 1 and _s.argv[1] == 'dma':      # register-staged vs LDS-DMA nine-tap kernel (DAT_WGRAD_DMA), twice each
    LAYERS[:] = LAYERS[:5]
    for rep in range(2):
        fresh({'DAT_WGRAD_DMA': '0'}, 'regs')
        fresh({'DAT_WGRAD_DMA': '1'}, 'dma')
elif len(_s.argv) > 1 and _s.argv[1] == 'sub':    # four-wave blocks (two per CU) vs eight-wave blocks with two K ranges (DAT_WGRAD_SUB), K-split sweep of the latter
    LAYERS[:] = LAYERS[:5]
    fresh({'DAT_WGRAD_SUB': '1'}, 'sub1')
    fresh({'DAT_WGRAD_SUB': '2'}, 'sub2')
    for ks in (4, 6, 8, 10, 12, 16, 20, 32, 42, 64):
        fresh({'DAT_WGRAD_SUB': '2', 'DAT_WGRAD_KS': str(ks)}, 'sub2ks%d' % ks)
    fresh({'DAT_WGRAD_SUB': '1'}, 'sub1')
    for name, rs in RESULTS.items():
        ref = rs[0][1]
        worst = max(((r - ref).abs().max() / ref.abs().max()).item() for _, r in rs[1:])
        print('%-28s worst max-abs difference to the first run / max-abs: %.3g' % (name, worst))
elif len(_s.argv) > 1 and _s.argv[1] == 'ilv':    # LDS-DMA pieces in one burst behind the barrier (0) vs between the MFMA groups (1)
    LAYERS[:] = LAYERS[:5]
    for m in (0, 1, 0, 1):
        fresh({'DAT_WGRAD_ILV': str(m)}, 'ilv%d' % m)
    for name, rs in RESULTS.items():
        ref = rs[0][1]
        worst = max(((r - ref).abs().max() / ref.abs().max()).item() for _, r in rs[1:])
        print('%-28s worst max-abs difference to the first run / max-abs: %.3g' % (name, worst))
elif len(_s.argv) > 1 and _s.argv[1] == 'res5':   # layers with more tiles than half the CUs: four-wave blocks (ks 4) vs eight-wave blocks with two / four K ranges
    LAYERS[:] = [LAYERS[2], ('kps head 3x3 512 (100 rois x 14 x 14)', 512, 512, (1, 3, 3), 1, 100, 14, 14, None)]
    for env, tag in (({'DAT_WGRAD_SUB': '1'}, 'four-wave'), ({'DAT_WGRAD_SUB': '2'}, 'sub2 ks2'), ({'DAT_WGRAD_SUB': '2', 'DAT_WGRAD_KS': '4'}, 'sub2 ks4'),
                     ({'DAT_WGRAD_SUB': '1'}, 'four-wave')):
        fresh(env, tag)
    for name, rs in RESULTS.items():
        ref = rs[0][1]
        worst = max(((r - ref).abs().max() / ref.abs().max()).item() for _, r in rs[1:])
        print('%-28s worst max-abs difference to the first run / max-abs: %.3g' % (name, worst))
elif len(_s.argv) > 1 and _s.argv[1] == 'pw':     # round 5: pointwise layers of an R-50 FPN3D training iteration -- 128 x 128 per-tap kernel vs wgrad_pw_kernel
    PW = [  # (res2 is frozen: no weight gradients below res3)
        ('res3_0 2a 256->128 s2', 256, 128, 2, 8, 192, 336), ('res3_0 sc 256->512 s2', 256, 512, 2, 8, 192, 336),
        ('res3 2c 128->512', 128, 512, 1, 8, 96, 168), ('res3 2a 512->128', 512, 128, 1, 8, 96, 168),
        ('res4_0 2a 512->256 s2', 512, 256, 2, 8, 96, 168), ('res4 2c 256->1024', 256, 1024, 1, 8, 48, 84), ('res4 2a 1024->256', 1024, 256, 1, 8, 48, 84),
        ('res5 2c 512->2048', 512, 2048, 1, 8, 24, 42), ('res5 2a 2048->512', 2048, 512, 1, 8, 24, 42),
        ('lateral P2 256->256 (3 frames)', 256, 256, 1, 3, 192, 336), ('lateral P3 512->256 (3 frames)', 512, 256, 1, 3, 96, 168),
        ('lateral P5 2048->256', 2048, 256, 1, 8, 24, 42), ('fc6 12544->1024 (512 rois)', 12544, 1024, 1, 1, 1, 512),
        ('rpn heads 256->15 P2 key frame', 256, 15, 1, 1, 192, 336),
    ]

    def run_pw(tag):
        for name, cin, cout, st, T, H, W in PW:
            g = torch.Generator().manual_seed(1)
            Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
            cs_g = ops.round_up(cout, 64)
            x = torch.randn((T, H, W, cin), generator=g).bfloat16().cuda()
            gy = torch.randn((T, Ho, Wo, cs_g), generator=g).bfloat16().cuda()
            gy[..., cout:] = 0
            w = torch.randn((cout, cin, 1, 1, 1), generator=g).cuda() * 0.05
            cg = ops.ConvGrad(w, None, (st, st), (0, 0, 0), ops.BF16, cin, cs_g)
            gt = torch.zeros(cout * cin, dtype=torch.float32, device='cuda')
            assert cg.weight_acc(x, gy, T, gt)
            torch.cuda.synchronize()
            RESULTS.setdefault(name, []).append((tag, gt.clone()))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                cg.weight_acc(x, gy, T, gt)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            fl = 2.0 * cin * cout * T * Ho * Wo
            byt = 2.0 * T * Ho * Wo * (cin + cs_g)
            print('%-10s %-34s %7.3f ms  %7.1f TFLOP/s  %6.2f TB/s compulsory' % (tag, name, ms, fl / ms / 1e9, byt / ms / 1e9), flush=True)

    def fresh_pw(env, tag):
        global run
        keep, run = run, run_pw
        try:
            fresh(env, tag)
        finally:
            run = keep
    def run_batch(tag):
        # the whole list as ONE dat_conv3d_wgrad_acc_batch call (what the training executor does per gradient bucket) against the sum of the
        # single launches
        jobs, fl = [], 0.0
        for name, cin, cout, st, T, H, W in PW:
            g = torch.Generator().manual_seed(1)
            Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
            cs_g = ops.round_up(cout, 64)
            x = torch.randn((T, H, W, cin), generator=g).bfloat16().cuda()
            gy = torch.randn((T, Ho, Wo, cs_g), generator=g).bfloat16().cuda()
            gy[..., cout:] = 0
            cg = ops.ConvGrad(torch.zeros((cout, cin, 1, 1, 1), device='cuda'), None, (st, st), (0, 0, 0), ops.BF16, cin, cs_g)
            gt = torch.zeros(cout * cin, dtype=torch.float32, device='cuda')
            jobs.append(cg.weight_acc_job(x, gy, T, gt))
            fl += 2.0 * cin * cout * T * Ho * Wo
        for n in (len(jobs), 8, 4):
            ops.wgrad_acc_batch(jobs[:n])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                for i in range(0, len(jobs), n):
                    ops.wgrad_acc_batch(jobs[i:i + n])
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print('%-10s all %d layers in batches of %2d: %7.3f ms  %7.1f TFLOP/s' % (tag, len(jobs), n, ms, fl / ms / 1e9), flush=True)

    def fresh_batch(env, tag):
        global run
        keep, run = run, run_batch
        try:
            fresh(env, tag)
        finally:
            run = keep
    fresh_batch({'DAT_WGRAD_PW': '1'}, 'batch')
    fresh_pw({'DAT_WGRAD_PW': '0'}, 'old128')
    fresh_pw({'DAT_WGRAD_PW': '1'}, 'pw')
    fresh_batch({'DAT_WGRAD_PW': '1'}, 'batch')
    if len(_s.argv) > 2 and _s.argv[2] == 'short':
        raise SystemExit(0)
    for shape in ('10', '20', '40'):
        fresh_pw({'DAT_WGRAD_PW': shape}, 'pw' + shape)
    for ks in (32, 64, 128, 512):
        fresh_pw({'DAT_WGRAD_PW': '1', 'DAT_WGRAD_KS': str(ks)}, 'pw ks%d' % ks)
    fresh_pw({'DAT_WGRAD_PW': '0'}, 'old128')
    for name, rs in RESULTS.items():
        ref = rs[0][1]
        worst = max(((r - ref).abs().max() / ref.abs().max()).item() for _, r in rs[1:])
        print('%-34s worst max-abs difference to the first run / max-abs: %.3g' % (name, worst))
elif len(_s.argv) > 1 and _s.argv[1] == 'ablate':
    LAYERS[:] = LAYERS[:5]
    for ab in (0, 1, 2, 4, 3, 5, 6, 7):
        fresh({'DAT_WGRAD_DIRECT': '1', 'DAT_WGRAD_ABLATE': str(ab)}, 'abl%d' % ab)
else:
    fresh({'DAT_WGRAD_DIRECT': '0'}, 'repack')
    fresh({'DAT_WGRAD_DIRECT': '1'}, 'direct')
    for ks in (4, 8, 16, 32, 64):
        fresh({'DAT_WGRAD_DIRECT': '1', 'DAT_WGRAD_KS': str(ks)}, 'ks%d' % ks)
