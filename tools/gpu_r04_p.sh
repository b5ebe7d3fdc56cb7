#!/bin/bash
# call P: proposal / NMS / detection kernels after the T = 1 register-resident instantiations (no scratch), model-level tests
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04p; mkdir -p $o
timeout 1500 python -m pytest -x -q -m gpu tests/test_gpu_kernels.py tests/test_gpu_model.py -k "nms or proposal or box_results or collect or batched or pipelin or forward or tube or c4 or detect or several" > $o/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $o/pytest.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-accuracy --h2d 0 > $o/bench.json 2>/dev/null; python -c "import json;d=json.load(open('$o/bench.json'));print(d['value'],d['ms_per_step'],d['sequential_clips_per_s'])"
