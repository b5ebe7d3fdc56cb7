#!/bin/bash
# round-2 final validation on one box: the whole -m gpu suite, smoke(), the driver's default bench line, the other BASELINE configs
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/${1:-r02final}; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q -x > $o/pytest_gpu.log 2>&1; echo "pytest -m gpu rc $?" | tee -a $o/pytest_gpu.log
tail -3 $o/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.log 2>&1; tail -1 $o/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc $?"
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy"
$B --pipeline 1 > $o/bench_p1.json 2> $o/bench_p1.err
$B --pipeline 1 --graph 0 > $o/bench_p1_g0.json 2> $o/bench_p1_g0.err
$B --graph 0 > $o/bench_g0.json 2> $o/bench_g0.err
$B --workload 3d_r50_fpn3d > $o/bench_r50.json 2> $o/bench_r50.err
$B --workload 2d_r50_fpn > $o/bench_2d.json 2> $o/bench_2d.err
$B --keyframe-dce > $o/bench_dce.json 2> $o/bench_dce.err
$B --mode train > $o/train18.json 2> $o/train18.err
$B --mode train --workload 3d_r50_fpn3d > $o/train50.json 2> $o/train50.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline',{})
        print(f.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d.get('sequential_clips_per_s'), r.get('kernel'), r.get('achieved'), r.get('frac'), r.get('all_conv_kernels',{}).get('ms_per_step'))
    except Exception as e: print(f,'ERR',e)
PY
