#!/bin/bash
# deferred weight-gradient finish: the training tests, then same-box A/B of bench.py --mode train (DAT_DEFER_WGRAD_FINISH=0 / 1)
tag=${1:-defer_check}
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/$tag; mkdir -p $o
cd $R
timeout -s KILL 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py -q -x -m gpu > $o/pytest.log 2>&1; grep -E "passed|failed|Error|error" $o/pytest.log | tail -4
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs --mode train --steps 20 --warmup 5"
for v in 0 1 0 1; do DAT_DEFER_WGRAD_FINISH=$v timeout -s KILL 300 $B 2>$o/err_$v.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEFER=$v r18 train ms', d['ms_per_step'])"; done
for v in 0 1; do DAT_DEFER_WGRAD_FINISH=$v timeout -s KILL 300 $B --workload 3d_r50_fpn3d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEFER=$v r50 train ms', d['ms_per_step'])"; done
