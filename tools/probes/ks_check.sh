#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 200 python tools/probes/wgrad_bench.py dma 2>&1 | grep -E "^dma" | head -5 | cut -c1-70
timeout -s KILL 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py -q -x -m gpu 2>&1 | grep -E "passed|failed" | tail -1
for i in 1 2; do timeout -s KILL 300 python bench.py --no-cpu-baseline --no-accuracy --no-other-configs --mode train --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r18 train ms', d['ms_per_step'])"; done
