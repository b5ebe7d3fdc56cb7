#!/bin/bash
# round-2 GPU call Y: dense stride-2 patch variant (NTAP=10): parity + per-layer A/B + bench
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02y; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "conv or full_size or forward" > $o/pytest_conv.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest_conv.log
tail -3 $o/pytest_conv.log
L="--arch R18 --iters 10 --cold --only res3_0_2a_s2,res4_0_2a_s2,res5_0_2a_s2"
run() { echo "$1: $(env $2 timeout 200 python tools/bench_layers.py $L 2>&1 | grep "res" | awk '{print $1, $(NF-3), $(NF-2), $(NF-1)}' | tr '\n' ';')"; }
run planes "DAT_CONV_NTAP=5"
run dense "X=1"
run planes "DAT_CONV_NTAP=5"
run dense "X=1"
B="timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy"
DAT_CONV_NTAP=5 $B > $o/bench_planes.json 2> $o/bench_planes.err
$B > $o/bench_dense.json 2> $o/bench_dense.err
DAT_CONV_NTAP=5 $B --workload 3d_r50_fpn3d > $o/bench50_planes.json 2> $o/bench50_planes.err
$B --workload 3d_r50_fpn3d > $o/bench50_dense.json 2> $o/bench50_dense.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline',{})
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'), r.get('achieved'), r.get('all_conv_kernels',{}).get('ms_per_step'))
    except Exception as e: print(f,'ERR',e)
PY
