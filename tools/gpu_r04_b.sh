#!/bin/bash
# round 4, call B: trainer re-ordering / exchange tests, the reworked bf16 4-clip gate, a short training bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04b
timeout 1500 python -m pytest -x -q -s -m gpu tests/test_gpu_train.py \
  "tests/test_gpu_parity_full.py::test_bf16_graph_of_four_clips_gives_every_clip_the_results_of_the_eager_one_clip_forward" \
  tests/test_gpu_train_full.py > gpurun_out/r04b/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04b/pytest.log
grep -E "passed|failed|rc=|clip [0-9]" gpurun_out/r04b/pytest.log | tail -40
timeout 300 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04b/train_r18.json 2> gpurun_out/r04b/train_r18.err
echo "train bench rc=$?"; head -c 600 gpurun_out/r04b/train_r18.json
