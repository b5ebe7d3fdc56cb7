#!/bin/bash
# LDS-DMA nine-tap weight gradient: the training tests (small shapes + the benched shape), then same-box A/B of bench.py --mode train
tag=${1:-wgrad_dma}
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/$tag; mkdir -p $o
cd $R
timeout -s KILL 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py -q -x -m gpu > $o/pytest.log 2>&1; grep -E "passed|failed" $o/pytest.log | tail -2
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs --mode train --steps 20 --warmup 5"
for v in 0 1 0 1; do DAT_WGRAD_DMA=$v timeout -s KILL 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DMA=$v r18 train ms', d['ms_per_step'])"; done
for v in 0 1; do DAT_WGRAD_DMA=$v timeout -s KILL 300 $B --workload 3d_r50_fpn3d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DMA=$v r50 train ms', d['ms_per_step'])"; done
