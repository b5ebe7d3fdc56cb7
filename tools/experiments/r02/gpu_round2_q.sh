#!/bin/bash
# round-2 GPU call Q: big-tile kernel: parity, per-layer timing, bench A/B
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02q; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "big_tile or conv3d" > $o/pytest_conv.log 2>&1; echo "pytest conv rc $?" | tee -a $o/pytest_conv.log
tail -5 $o/pytest_conv.log
for v in 0 1; do
  DAT_CONV_BT=$v timeout 200 python tools/bench_layers.py --arch R18 --iters 10 > $o/layers_bt$v.log 2>&1
  grep "fpn_post_P2\|fpn_post_P3\|TOTAL" $o/layers_bt$v.log
done
B="timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy"
DAT_CONV_BT=0 $B > $o/bench_bt0.json 2> $o/bench_bt0.err
$B > $o/bench_bt1.json 2> $o/bench_bt1.err
DAT_CONV_BT=0 $B --pipeline 1 > $o/bench_bt0_p1.json 2> $o/bench_bt0_p1.err
$B --pipeline 1 > $o/bench_bt1_p1.json 2> $o/bench_bt1_p1.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'), d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['all_conv_kernels']['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
