#!/bin/bash
# round 4, call E: fewer launches per forward (no pad fills, lazy key-frame slices, packed zero inits, resident graph inputs)
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04e; mkdir -p $o
timeout 1500 python -m pytest -x -q -m gpu tests/test_gpu_model.py tests/test_gpu_parity_full.py -k "not bf16x3" > $o/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $o/pytest.log
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs > $o/bench.json 2> $o/bench.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04e/bench.json'))
print(d['value'], d['ms_per_step'], d['sequential_clips_per_s'], d['host_frames']['value_including_upload'], d['roofline']['frac'], d['roofline']['all_conv_kernels'])
PY
tail -3 $o/bench.err
