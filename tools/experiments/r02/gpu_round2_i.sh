#!/bin/bash
# round-2 GPU call I: full suite (WD=2 default, tiled pack fix, ClipGraph), bench with / without hipGraph, training
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02i; mkdir -p $o
python -m pytest tests -m gpu -q > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log
tail -3 $o/pytest.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy"
for g in 1 0; do
  $B --graph $g > $o/bench_g$g.json 2> $o/bench_g$g.err
  $B --graph $g --pipeline 1 > $o/bench_g${g}_p1.json 2> $o/bench_g${g}_p1.err
  $B --graph $g --pipeline 2 > $o/bench_g${g}_p2.json 2> $o/bench_g${g}_p2.err
done
$B --graph 1 --pipeline 4 > $o/bench_g1_p4.json 2> $o/bench_g1_p4.err
python bench.py --steps 5 --warmup 2 --workload 2d_r50_fpn --no-cpu-baseline --no-accuracy > $o/bench_2d.json 2> $o/bench_2d.err
python bench.py --steps 5 --warmup 2 --arch 50 --no-cpu-baseline --no-accuracy > $o/bench_r50.json 2> $o/bench_r50.err
python bench.py --steps 8 --warmup 3 --mode train --no-cpu-baseline > $o/train18.json 2> $o/train18.err
python bench.py --steps 8 --warmup 3 --mode train --arch 50 --no-cpu-baseline > $o/train50.json 2> $o/train50.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'), d.get('host_enqueue_ms_per_step'), d['config'].get('hip_graph'), d['roofline']['achieved'], d['roofline']['all_conv_kernels']['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
grep -h "capture failed" $o/*.err | head -3
