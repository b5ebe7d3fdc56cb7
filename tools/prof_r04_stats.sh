#!/bin/bash
# the --pipeline 1 kernel-stats pass of tools/prof_r04.sh alone (refresh after changes outside the conv kernels) -> gpurun_out/$1/
tag=${1:-r04stats}
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs --h2d 0"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o r1 -- $B --steps 10 --warmup 3 --pipeline 1 > $o/stats.log 2>&1
for g in $o/stats/*/r1_*.csv; do [ -f "$g" ] && mv $g $o/stats/; done
rm -f $o/stats/r1_kernel_trace.csv
grep -h '"metric"' $o/stats.log | cut -c1-160
head -12 $o/stats/r1_kernel_stats.csv | cut -c1-150
