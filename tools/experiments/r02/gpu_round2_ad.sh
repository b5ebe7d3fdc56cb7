#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r02ad; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o r1 -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $o/stats.log 2>&1
head -40 $o/stats/r1_kernel_stats.csv | cut -c1-160
