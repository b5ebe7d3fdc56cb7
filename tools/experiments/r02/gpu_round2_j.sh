#!/bin/bash
# round-2 GPU call J: bench with / without hipGraph at several pipeline depths, 2D / R-50 workloads, plan-model check
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02j; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "nms or box_results or two_contexts" > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log
B="timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy"
for p in 1 2 3 4; do
  $B --graph 1 --pipeline $p > $o/bench_g1_p$p.json 2> $o/bench_g1_p$p.err
done
$B --graph 0 --pipeline 3 > $o/bench_g0_p3.json 2> $o/bench_g0_p3.err
timeout 200 python bench.py --steps 5 --warmup 2 --workload 2d_r50_fpn --no-cpu-baseline --no-accuracy > $o/bench_2d.json 2> $o/bench_2d.err
timeout 200 python bench.py --steps 5 --warmup 2 --workload 2d_r50_fpn --no-cpu-baseline --no-accuracy --graph 0 > $o/bench_2d_g0.json 2> $o/bench_2d_g0.err
timeout 200 python bench.py --steps 5 --warmup 2 --arch 50 --no-cpu-baseline --no-accuracy > $o/bench_r50.json 2> $o/bench_r50.err
python tools/tune_plan.py --arch R18 --iters 6 > $o/tune_r18.log 2>&1
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'), d.get('host_enqueue_ms_per_step'), d['config'].get('hip_graph'), d['roofline']['achieved'], d['roofline']['all_conv_kernels']['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
grep -h "capture failed\|fault" $o/*.err | head -3; tail -4 $o/tune_r18.log
