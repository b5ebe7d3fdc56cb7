"""A/B of dat_kps_finalize: the per-(roi, frame) tile kernel against the per-element kernel (DAT_KPS_FINALIZE_TILE=0) -- run once per
setting with the same seed; prints the time per launch and saves the output for a bit-for-bit comparison."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detectandtrack_amd.ops import hip_ops as ops  # noqa: E402

out_path = sys.argv[1]
g = torch.Generator(device='cuda').manual_seed(5)
res = {}
for name, dt, tdt in (('bf16', ops.BF16, torch.bfloat16), ('fp32', ops.F32, torch.float32)):
    for (R, Tr, K) in ((400, 1, 17), (37, 3, 17), (5, 1, 6)):
        sub = torch.randn((R * Tr, 14, 14, 128), device='cuda', generator=g).to(tdt)
        y = ops.kps_finalize(sub, dt, R, Tr, K, 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.kps_finalize(sub, dt, R, Tr, K, 2)
        e1.record()
        torch.cuda.synchronize()
        print('%s R=%d Tr=%d K=%d: %.1f us' % (name, R, Tr, K, e0.elapsed_time(e1) / 20 * 1e3))
        res['%s_%d_%d_%d' % (name, R, Tr, K)] = y.cpu().numpy()
np.savez(out_path, **res)
