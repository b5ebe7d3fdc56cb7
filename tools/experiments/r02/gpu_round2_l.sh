#!/bin/bash
# round-2 GPU call L: conv1 + pool1 fused kernel: parity, model tests, A/B in the bench
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02l; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "stem" > $o/pytest_stem.log 2>&1; echo "pytest stem rc $?" | tee -a $o/pytest_stem.log
tail -3 $o/pytest_stem.log
python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_full.py tests/test_gpu_train.py -m gpu -q -x > $o/pytest_model.log 2>&1; echo "pytest model rc $?" | tee -a $o/pytest_model.log
tail -3 $o/pytest_model.log
B="timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy"
for rep in 1 2; do
DAT_FUSE_STEM_POOL=0 $B > $o/bench_unfused.$rep.json 2> $o/bench_unfused.$rep.err
$B > $o/bench_fused.$rep.json 2> $o/bench_fused.$rep.err
done
DAT_FUSE_STEM_POOL=0 $B --pipeline 1 > $o/bench_unfused_p1.json 2> $o/bench_unfused_p1.err
$B --pipeline 1 > $o/bench_fused_p1.json 2> $o/bench_fused_p1.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'), d['roofline']['achieved'], d['roofline']['all_conv_kernels']['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
