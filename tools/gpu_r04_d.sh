#!/bin/bash
# round 4, call D: evidence -- rocprofv3 kernel stats of the headline with ONE forward in flight (graph replay), and the ordered kernel
# trace of eager forwards (which launches of a forward are not ours)
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r04d; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs --h2d 0"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_p1 -o r1 -- $B --steps 10 --warmup 3 --pipeline 1 > $o/stats_p1.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $o/trace_eager -o r1 -- $B --steps 2 --warmup 1 --pipeline 1 --graph 0 > $o/trace_eager.log 2>&1
for d in stats_p1 trace_eager; do for g in $o/$d/*/r1_*.csv; do [ -f "$g" ] && mv $g $o/$d/; done; done
ls -la $o/stats_p1 $o/trace_eager | head -20
tail -2 $o/stats_p1.log | cut -c1-300
# keep the pulled files small: the trace as (start, name) only
python - <<'PY'
import csv, os
o = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r04d'
src = o + '/trace_eager/r1_kernel_trace.csv'
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
with open(o + '/trace_eager_seq.txt', 'w') as f:
    for r in rows:
        f.write('%s %s %s\n' % (r['Start_Timestamp'], int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'][:120]))
os.remove(src)
PY
