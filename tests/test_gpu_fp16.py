"""The fp16-operand performance mode (round 6): libdat_hip_f16.so = the same sources compiled with -DDAT_H16_IS_FP16 (IEEE
half in every 16-bit tensor and packed weight, v_mfma_f32_32x32x16_f16).  The format belongs to the loaded library, so the mode runs in a
process of its own (DAT_H16=fp16).  Gate: against the oracle on the small R-18 clip the fp16 blobs must be an order of magnitude closer than
the bf16 mode's 6 % gate (tests/test_gpu_model.py::test_bf16_forward_close_to_oracle) -- 1 % of range on every checked blob, kps_score
within 2e-2 -- and a workspace in the OTHER 16-bit mode than the loaded build must be refused."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fp16_mode_forward_is_close_to_the_oracle():
    env = dict(os.environ, DAT_H16='fp16', PYTHONPATH=REPO)
    p = subprocess.run([sys.executable, os.path.join(REPO, 'tests', 'fp16_forward.py')], env=env, cwd=REPO, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = json.loads(p.stdout.decode().strip().splitlines()[-1])
    print(out)
    assert out['h16_format'] == 1 and out['tensor_dtype'] == 'torch.float16'
    for n, rel in out['blobs'].items():
        assert rel < 0.01, (n, rel)
    assert out['rois'][0] == out['rois'][1] and out['rois_found_within_1px'] > 0.9
    assert out['kps_max_abs_err'] < 2e-2 * max(1.0, out['kps_ref_max_abs']) and out['kps_argmax_identical'] > 0.97
    assert out['x3_refused'] is True


def test_the_16_bit_mode_must_match_the_loaded_build():
    """This process holds the bf16 build: cfg.HIP.DTYPE 'fp16' is refused with the instruction to start the process with DAT_H16=fp16."""
    from tests.model_util import fpn3d_kps_cfg, build_product, synthetic_clip
    import numpy as np
    model, ws, _ = build_product(fpn3d_kps_cfg('18', T=2, dtype='fp16'))
    ws.FeedBlob('data', synthetic_clip(2, 64, 96))
    ws.FeedBlob('im_info', np.array([[64, 96, 1.0]], dtype=np.float32))
    with pytest.raises(AssertionError, match='DAT_H16=fp16'):
        ws.RunNet(model.net.name)
