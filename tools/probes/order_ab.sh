#!/bin/bash
# A/B of the conv block order inside an XCD's queue (DAT_CONV_ORDER: 0 patch-sharing, 1 weight-stationary, -1 rule):
# per-layer table, FETCH_SIZE pass and the bench value of each -> gpurun_out/$1/
tag=${1:-order_ab}
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs"
for ord in 0 1 -1; do
  export DAT_CONV_ORDER=$ord
  timeout -s KILL 300 $B --steps 10 --warmup 3 --pipeline 1 --graph 0 --h2d 0 --dump-convs > $o/seq_$ord.json 2> $o/conv_layers_$ord.txt
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/fetch_$ord -o r1 -- $B --steps 3 --warmup 1 --pipeline 1 --h2d 0 > $o/fetch_$ord.log 2>&1
  for g in $o/fetch_$ord/*/r1_*.csv; do [ -f "$g" ] && mv $g $o/fetch_$ord/; done
done
for ord in 0 -1 0 -1; do DAT_CONV_ORDER=$ord timeout -s KILL 300 $B --steps 40 --warmup 5 --h2d 0 > $o/bench_${ord}_$RANDOM.json 2>> $o/bench.err; done
ls $o
