import numpy as np, sys
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
from model_util import build_product, fpn3d_kps_cfg
from detectandtrack_amd.core import test as test_engine
from detectandtrack_amd.core.config import cfg
T = 2
model, ws, _ = build_product(fpn3d_kps_cfg('18', T=T, dtype='fp32'))
cfg.TEST.SCALES = (64,); cfg.TEST.MAX_SIZE = 128; cfg.TEST.SCORE_THRESH = 0.0
rs = np.random.RandomState(0)
frames = [rs.randint(0, 255, (60, 90, 3)).astype(np.uint8) for _ in range(T)]
a = test_engine.im_detect_all(model, frames, None)
b = test_engine.im_detect_all(model, frames, None)
print('device path twice: boxes equal', np.array_equal(a[0][1], b[0][1]), 'keyps equal', all(np.array_equal(x, y) for x, y in zip(a[2][1], b[2][1])))
cfg.HIP.DEVICE_BOX_RESULTS = False
c = test_engine.im_detect_all(model, frames, None)
d = test_engine.im_detect_all(model, frames, None)
print('host glue twice: keyps equal', all(np.array_equal(x, y) for x, y in zip(c[2][1], d[2][1])))
print('box max diff', np.abs(a[0][1] - c[0][1]).max())
for i, (x, y) in enumerate(zip(a[2][1], c[2][1])):
    bad = np.argwhere(np.abs(x[:2] - y[:2]) > 5e-3)
    for r, k in bad:
        print('det', i, 'row', r, 'kp', k, 'coords', x[:2, k], y[:2, k], 'logit', x[2, k], y[2, k], 'prob', x[3, k], y[3, k])
