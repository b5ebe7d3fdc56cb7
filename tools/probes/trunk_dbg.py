"""Debug: pipelined engine, plain vs frame-trunk cache, graph vs eager launches -- which clips differ between which modes."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from tests.model_util import fpn3d_kps_cfg
from detectandtrack_amd.core import test_engine
from detectandtrack_amd.core.config import cfg, cfg_from_cfg, assert_and_infer_cfg, reset_cfg
from detectandtrack_amd import workspace

T, H, W, n_frames = 4, 96, 128, 7
rs = np.random.RandomState(9)
roidb = []
for v in range(2):
    video = [rs.randint(0, 255, (H, W, 3)).astype(np.uint8) for _ in range(n_frames)]
    for k in range(n_frames):
        ids = [min(max(k - T // 2 + j, 0), n_frames - 1) for j in range(T)]
        roidb.append({'image': [video[i] for i in ids], 'frame_ids': [('vid%d' % v, i) for i in ids], 'height': H, 'width': W})


def run(cache, per, graph, depth):
    c = fpn3d_kps_cfg('18', T=T, dtype='fp32', pre=300, post=100)
    c['TEST'].update(SCALES=(H,), MAX_SIZE=max(H, W), SCORE_THRESH=0.0, DETECTIONS_PER_IM=15)
    c['HIP'].update(PIPELINE_DEPTH=depth, IMS_PER_FORWARD=per, CLIP_GRAPH=graph, FRAME_TRUNK_CACHE=cache)
    c['RNG_SEED'] = 3
    reset_cfg(); cfg_from_cfg(c); assert_and_infer_cfg()
    workspace.ResetWorkspace()
    return test_engine.test_net(roidb, None, None)['all_boxes'][1]


if len(sys.argv) > 2 and sys.argv[1] == 'loop':
    # the intermittent eager-launch mismatch: one graph reference, then N x (plain eager, cached eager) -- which side differs, where, by how much
    ref = run(0, 2, True, 3)
    for it in range(int(sys.argv[2])):
        for name, args in (('plain e per2', (0, 2, False, 3)), ('cache e per2', (10, 2, False, 3))):
            got = run(*args)
            for k in range(len(roidb)):
                a, b = got[k], ref[k]
                if a.shape != b.shape or not np.array_equal(a, b):
                    d = np.abs(a - b).max() if a.shape == b.shape else float('nan')
                    rows = int((a != b).any(axis=1).sum()) if a.shape == b.shape else -1
                    print('iter %d %s: clip %d differs (shapes %s %s, %d rows, max |d| %g)' % (it, name, k, a.shape, b.shape, rows, d), flush=True)
        print('iter %d done' % it, flush=True)
    sys.exit(0)

seq = [('plain g per1', (0, 1, True, 3)), ('cache g per1', (6, 1, True, 3)), ('plain g per2', (0, 2, True, 3)), ('cache g per2', (10, 2, True, 3)),
       ('plain e per2', (0, 2, False, 3)), ('cache e per2', (10, 2, False, 3)), ('plain e per2 again', (0, 2, False, 3)), ('cache e per2 again', (10, 2, False, 3))]
if len(sys.argv) > 1:
    seq = seq[int(sys.argv[1]):]
runs = {}
for name, args in seq:
    runs[name] = run(*args)
    ref = runs.get('plain e per2', runs[seq[0][0]])
names = list(runs)
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        a, b = runs[names[i]], runs[names[j]]
        bad = [k for k in range(len(roidb)) if a[k].shape != b[k].shape or not np.array_equal(a[k], b[k])]
        if 'per1' in names[i] and 'per2' in names[j]:
            continue
        print('%-20s vs %-20s differing clips: %s' % (names[i], names[j], bad))
