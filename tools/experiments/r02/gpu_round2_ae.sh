#!/bin/bash
# round-2 GPU call AE: batched weight re-pack: parity, training tests, training bench
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02ae; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py -m gpu -q -x -k "repack or train or step or sgd or loss" > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log
tail -3 $o/pytest.log
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --mode train"
$B > $o/train18.json 2> $o/train18.err
$B --workload 3d_r50_fpn3d > $o/train50.json 2> $o/train50.err
$B > $o/train18b.json 2> $o/train18b.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
