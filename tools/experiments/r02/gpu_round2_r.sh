#!/bin/bash
# round-2 GPU call R: big-tile kernel on evenly filling grids only (P2): tests + bench A/B, two repetitions
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02r; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "big_tile or weights_stationary" > $o/pytest_conv.log 2>&1; echo "pytest conv rc $?" | tee -a $o/pytest_conv.log
tail -3 $o/pytest_conv.log
B="timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy"
for rep in 1 2; do
DAT_CONV_BT=0 $B > $o/bench_bt0.$rep.json 2> $o/bench_bt0.$rep.err
$B > $o/bench_bt1.$rep.json 2> $o/bench_bt1.$rep.err
done
DAT_CONV_BT=0 $B --pipeline 1 > $o/bench_bt0_p1.json 2> $o/bench_bt0_p1.err
$B --pipeline 1 > $o/bench_bt1_p1.json 2> $o/bench_bt1_p1.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'), d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['all_conv_kernels']['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
