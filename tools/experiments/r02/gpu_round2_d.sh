#!/bin/bash
# round-2 GPU call D: full GPU suite + bench with the device-side glue + rocprofv3 kernel stats of the default bench
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02d; mkdir -p $o
python -m pytest tests -m gpu -q -s > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log
grep -E "passed|failed" $o/pytest.log | tail -2
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-convs > $o/bench.json 2> $o/bench.err; echo "bench rc $?"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --pipeline 1 > $o/bench_p1.json 2> $o/bench_p1.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --pipeline 2 > $o/bench_p2.json 2> $o/bench_p2.err
python bench.py --steps 5 --warmup 2 --workload 2d_r50_fpn --no-cpu-baseline --no-accuracy > $o/bench_2d.json 2> $o/bench_2d.err
python bench.py --steps 5 --warmup 2 --arch 50 --no-cpu-baseline --no-accuracy > $o/bench_r50.json 2> $o/bench_r50.err
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
timeout -s KILL 250 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o r1 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy > $o/stats.log 2>&1
cd $R
python - <<PY
import json
for f in ('bench','bench_p1','bench_p2','bench_2d','bench_r50'):
    try:
        d=json.load(open('$o/%s.json'%f)); r=d['roofline']; print(f, d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'), r['kernel'], r['achieved'], r['avg_launch_ms'], r['all_conv_kernels'])
    except Exception as e: print(f,'ERR',e)
PY
