#!/bin/bash
# round-2 GPU call U: weights-stationary 1x1 64->256 kernel: parity + cold-cache timing + bench
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02u; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv" > $o/pytest_conv.log 2>&1; echo "pytest conv rc $?" | tee -a $o/pytest_conv.log
tail -3 $o/pytest_conv.log
L="--arch R18 --iters 10 --cold --topdown --only fpn_lat_P2"
run() { echo "$1: $(env $2 timeout 200 python tools/bench_layers.py $L 2>&1 | grep "fpn_" | awk '{print $1, $(NF-3), $(NF-2)}' | tr '\n' ';')"; }
run generic "DAT_CONV_WS64=0"
run pw256 "X=1"
run pw256_nostores "DAT_CONV_ABLATE=4"
B="timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy"
$B > $o/bench.json 2> $o/bench.err
$B --pipeline 1 > $o/bench_p1.json 2> $o/bench_p1.err
$B --pipeline 1 --graph 0 --dump-convs > $o/bench_dump.json 2> $o/bench_dump.err
grep "fpn_inner_res" $o/bench_dump.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'), d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['all_conv_kernels']['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
