#!/bin/bash
# call R: dat_kps_finalize tile kernel vs the per-element kernel: time and bit-identity; the keypoint tests
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04r; mkdir -p $o
echo "tile kernel:"; python tools/probes/kps_finalize_ab.py /tmp/kf_tile.npz 2>/dev/null
echo "per-element kernel:"; DAT_KPS_FINALIZE_TILE=0 python tools/probes/kps_finalize_ab.py /tmp/kf_elem.npz 2>/dev/null
python - <<'PY'
import numpy as np
a, b = np.load('/tmp/kf_tile.npz'), np.load('/tmp/kf_elem.npz')
for k in a.files:
    print(k, 'bit-identical' if np.array_equal(a[k], b[k]) else 'DIFFERENT max %.3e' % np.abs(a[k] - b[k]).max())
PY
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_kernels.py tests/test_gpu_model.py -k "kps or keypoint or heatmap or forward_matches" > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest.log
