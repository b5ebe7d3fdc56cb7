#!/bin/bash
# round-2 GPU call V: residual rows fetched a position group ahead (all conv epilogues): parity + cold-cache timing + bench
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02v; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv" > $o/pytest_conv.log 2>&1; echo "pytest conv rc $?" | tee -a $o/pytest_conv.log
tail -3 $o/pytest_conv.log
L="--arch R18 --iters 10 --cold --topdown --only fpn_lat_P2,fpn_lat_P3,fpn_lat_P4"
run() { echo "$1: $(env $2 timeout 200 python tools/bench_layers.py $L 2>&1 | grep "fpn_" | awk '{print $1, $(NF-3), $(NF-2)}' | tr '\n' ';')"; }
run generic "DAT_CONV_WS64=0"
run pw256 "X=1"
B="timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy"
DAT_CONV_WS64=0 $B --pipeline 1 --graph 0 --dump-convs > $o/bench_dump_gen.json 2> $o/bench_dump_gen.err
$B --pipeline 1 --graph 0 --dump-convs > $o/bench_dump.json 2> $o/bench_dump.err
echo "generic 1x1 / no ws64:"; grep "fpn_inner_res\|res2_1_sum\|res3_1_sum\|res4_1_sum\|res5_1_sum" $o/bench_dump_gen.err
echo "default:"; grep "fpn_inner_res\|res2_1_sum\|res3_1_sum\|res4_1_sum\|res5_1_sum" $o/bench_dump.err
$B > $o/bench.json 2> $o/bench.err
$B --pipeline 1 > $o/bench_p1.json 2> $o/bench_p1.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'), d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['all_conv_kernels']['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
