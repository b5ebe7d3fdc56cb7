"""A/B of dat_heatmaps_to_keypoints: the separable kernel (default) against the per-pixel 4 x 4 kernel (DAT_KPS_DECODE_SEP=0) at the bench's
size (400 rois x 17 maps of 56 x 56, boxes as the bench's detections have them) and on small / huge / degenerate boxes: time per launch, and
x / y / logit must be BIT-identical (the probability to the summation order)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from detectandtrack_amd.ops import hip_ops as ops  # noqa: E402

rs = np.random.RandomState(3)
cases = []
for name, R, T, K, wlo, whi in (('bench 400 rois', 400, 1, 17, 40, 500), ('small boxes', 64, 1, 17, 1, 30), ('huge boxes', 16, 1, 17, 700, 1279),
                                ('tubes', 40, 3, 17, 20, 300), ('one column', 8, 1, 5, 0.2, 1.0)):
    maps = (rs.randn(R, T * K, 56, 56) * 2).astype(np.float32)
    x1, y1 = rs.uniform(0, 600, (R, T)), rs.uniform(0, 300, (R, T))
    w, h = rs.uniform(wlo, whi, (R, T)), rs.uniform(wlo, whi, (R, T)) * 0.7
    boxes = np.stack([x1, y1, x1 + w, y1 + h], axis=2).reshape(R, 4 * T).astype(np.float32)
    cases.append((name, torch.from_numpy(maps).cuda(), torch.from_numpy(boxes).cuda(), T, K))
res = {}
for mode in ('0', '1', '0', '1'):
    os.environ['DAT_KPS_DECODE_SEP'] = mode
    with torch.cuda.stream(torch.cuda.Stream()):
        ops.drop_ctx()
        for name, maps, boxes, T, K in cases:
            out = ops.heatmaps_to_keypoints(maps, boxes, T, K)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            for _ in range(10):
                ops.heatmaps_to_keypoints(maps, boxes, T, K)
            e1.record(torch.cuda.current_stream())
            torch.cuda.synchronize()
            print('sep=%s %-16s %8.1f us' % (mode, name, e0.elapsed_time(e1) / 10 * 1e3), flush=True)
            res.setdefault(name, {})[mode] = out.cpu().numpy()
        ops.drop_ctx()
ok = True
for name, r in res.items():
    a, b = r['0'], r['1']
    same = np.array_equal(a[:, :3], b[:, :3])
    dp = float(np.abs(a[:, 3] - b[:, 3]).max() / max(a[:, 3].max(), 1e-30))
    print('%-16s x / y / logit bit-identical: %s; prob rel. difference %.2e' % (name, same, dp))
    ok = ok and same and dp < 1e-4
print('OK' if ok else 'MISMATCH')
sys.exit(0 if ok else 1)
