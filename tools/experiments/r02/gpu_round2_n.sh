#!/bin/bash
# round-2 GPU call N: weights-stationary 3x3 64->64 kernel: parity, per-layer timing, bench A/B
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02n; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv" > $o/pytest_conv.log 2>&1; echo "pytest conv rc $?" | tee -a $o/pytest_conv.log
tail -5 $o/pytest_conv.log
for v in 0 1; do
  DAT_CONV_WS64=$v timeout 200 python tools/bench_layers.py --arch R18 --iters 10 > $o/layers_ws$v.log 2>&1
  grep "res2_3x3\|TOTAL" $o/layers_ws$v.log
done
B="timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy"
DAT_CONV_WS64=0 $B > $o/bench_ws0.json 2> $o/bench_ws0.err
$B > $o/bench_ws1.json 2> $o/bench_ws1.err
DAT_CONV_WS64=0 $B --pipeline 1 > $o/bench_ws0_p1.json 2> $o/bench_ws0_p1.err
$B --pipeline 1 > $o/bench_ws1_p1.json 2> $o/bench_ws1_p1.err
$B --pipeline 1 --graph 0 --dump-convs > $o/bench_dump.json 2> $o/bench_dump.err
grep "res2_\|fpn_inner_res2" $o/bench_dump.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'), d['roofline']['achieved'], d['roofline']['all_conv_kernels']['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
