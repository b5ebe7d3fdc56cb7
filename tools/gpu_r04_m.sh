#!/bin/bash
# call M: bf16x3 with the split written by the producing conv's epilogue: kernel test, full-shape parity, throughput
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04m; mkdir -p $o
timeout 900 python -m pytest -x -q -m gpu "tests/test_gpu_kernels.py::test_conv3d" -k "bf16x3" > $o/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $o/pytest_kernels.log
timeout 1500 python -m pytest -x -q -s -m gpu tests/test_gpu_parity_full.py -k "bf16x3 and (four_clips or 18)" > $o/pytest_parity.log 2>&1; echo "parity rc=$?"; grep -E "passed|failed|kps_score" $o/pytest_parity.log | tail -8
timeout 300 python bench.py --dtype bf16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --no-other-configs --h2d 0 > $o/bench_x3.json 2> $o/bench_x3.err
echo "bench rc=$?"; python -c "import json;d=json.load(open('$o/bench_x3.json'));print('x3',d['value'],d['ms_per_step'],d['sequential_clips_per_s'],d['roofline']['all_conv_kernels'])"
