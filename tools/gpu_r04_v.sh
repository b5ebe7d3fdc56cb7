#!/bin/bash
# call V: dat_preprocess_frames with four pixels per thread: bit-identity tests, the host-frame leg of the bench
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04v; mkdir -p $o
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_kernels.py tests/test_gpu_model.py -k "preprocess or pipelined" > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest.log
python - <<'PY'
import torch, time
from detectandtrack_amd.ops import hip_ops as ops
from detectandtrack_amd.core.config import cfg
fr = torch.randint(0, 255, (32, 720, 1280, 3), dtype=torch.uint8, device='cuda')
sc = min(800 / 720.0, 1333 / 1280.0)
d, _ = ops.preprocess_frames(fr, 8, sc, cfg.PIXEL_MEANS, 32)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.preprocess_frames(fr, 8, sc, cfg.PIXEL_MEANS, 32, out=d)
e1.record(); torch.cuda.synchronize()
print('preprocess 32 frames 720x1280 -> %s: %.1f us' % (tuple(d.shape), e0.elapsed_time(e1) / 20 * 1e3))
PY
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-accuracy > $o/bench.json 2>/dev/null; python -c "import json;d=json.load(open('$o/bench.json'));print('value',d['value'],'host frames',d['host_frames']['value_including_upload'], d['host_frames']['ms_per_step'], d['ms_per_step'])"
