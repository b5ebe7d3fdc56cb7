#!/bin/bash
# round-4 profiling of the training iterations (after the eight-wave weight-gradient blocks) -> gpurun_out/$1/
#   train_r18 / train_r50 kernel stats (rocprofv3 --kernel-trace --stats); pmc_mfma_train: MFMA counters of the R-18 iteration (--kernel-trace only)
tag=${1:-r04train}
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
prof() { n=$1; shift
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/$n -o r1 -- python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs --steps 10 --warmup 3 "$@" > $o/$n.log 2>&1
    f=$(ls $o/$n/*/r1_kernel_stats.csv $o/$n/r1_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $o/${n}_kernel_stats.csv; rm -rf $o/$n; }
prof train_r18 --mode train
prof train_r50 --mode train --workload 3d_r50_fpn3d
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/pmc_mfma -o r1 -- python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs --mode train --steps 3 --warmup 1 > $o/pmc_mfma.log 2>&1
for g in $o/pmc_mfma/*/r1_*.csv; do [ -f "$g" ] && mv $g $o/pmc_mfma/; done
grep -h '"metric"' $o/train_r18.log $o/train_r50.log | cut -c1-200
ls -la $o $o/pmc_mfma | head -30
