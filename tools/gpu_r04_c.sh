#!/bin/bash
# round 4, call C: the bf16x3 mode -- kernel tests, full-shape oracle parity, throughput next to fp32
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04c
timeout 1200 python -m pytest -x -q -s -m gpu "tests/test_gpu_kernels.py::test_conv3d" -k "bf16x3" > gpurun_out/r04c/pytest_kernels.log 2>&1
echo "kernels rc=$?"; grep -E "passed|failed|err " gpurun_out/r04c/pytest_kernels.log | tail -30
timeout 1500 python -m pytest -x -q -s -m gpu tests/test_gpu_parity_full.py -k "bf16x3" > gpurun_out/r04c/pytest_parity.log 2>&1
echo "parity rc=$?"; grep -E "passed|failed|max-abs|rois:|Error" gpurun_out/r04c/pytest_parity.log | tail -60
timeout 300 python bench.py --dtype bf16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --no-other-configs --h2d 0 --dump-convs > gpurun_out/r04c/bench_x3.json 2> gpurun_out/r04c/bench_x3.err
echo "bench rc=$?"; head -c 400 gpurun_out/r04c/bench_x3.json; echo; grep -E "ms/step" gpurun_out/r04c/bench_x3.err | head -40
