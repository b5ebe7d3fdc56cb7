#!/bin/bash
# weight re-pack kernel after a change: the tests that consume packed weights, then its duration in a short training profile
tag=${1:-pack_probe}
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/$tag; mkdir -p $o
cd $R
timeout -s KILL 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py -q -x -m gpu -k "conv3d or repack or train or grad or wgrad or dgrad" > $o/pytest.log 2>&1; tail -3 $o/pytest.log
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/train -o r1 -- $B --steps 10 --warmup 3 --mode train > $o/train.log 2>&1
f=$(ls $o/train/*/r1_kernel_stats.csv $o/train/r1_kernel_stats.csv 2>/dev/null | head -1); grep -E "pack_weights|Name" $f | cut -c1-200
for i in 1 2; do timeout -s KILL 300 $B --steps 20 --warmup 5 --mode train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train ms', d['ms_per_step'])"; done
