#!/bin/bash
# round-2 GPU call AA: linear position tiling of small maps: parity + head-layer timing + bench A/B
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02aa; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_parity_full.py -m gpu -q -x -k "conv or forward or oracle" > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log
tail -3 $o/pytest.log
B="timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy"
for rep in 1 2; do
DAT_CONV_LINEAR=0 $B > $o/bench_lin0.$rep.json 2> $o/bench_lin0.$rep.err
$B > $o/bench_lin1.$rep.json 2> $o/bench_lin1.$rep.err
done
DAT_CONV_LINEAR=0 $B --pipeline 1 --graph 0 --dump-convs > $o/dump_lin0.json 2> $o/dump_lin0.err
$B --pipeline 1 --graph 0 --dump-convs > $o/dump_lin1.json 2> $o/dump_lin1.err
echo lin0; grep "conv_fcn\|conv_rpn" $o/dump_lin0.err | head -12
echo lin1; grep "conv_fcn\|conv_rpn" $o/dump_lin1.err | head -12
DAT_CONV_LINEAR=0 $B --workload 3d_r50_fpn3d > $o/bench50_lin0.json 2> $o/bench50_lin0.err
$B --workload 3d_r50_fpn3d > $o/bench50_lin1.json 2> $o/bench50_lin1.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline',{})
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'), r.get('achieved'), r.get('all_conv_kernels',{}).get('ms_per_step'))
    except Exception as e: print(f,'ERR',e)
PY
