#!/bin/bash
# round-2 profiling passes of the default bench workload -> gpurun_out/$1/ (then: python tools/pmc_summary.py gpurun_out/$1 profiles/$1)
#   stats      rocprofv3 --kernel-trace --stats         bench.py default (3 clips in flight)
#   pmc_mfma   --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE    (--pipeline 1)
#   pmc_fetch  --pmc FETCH_SIZE ; pmc_write --pmc WRITE_SIZE   (separate passes, --pipeline 1)
# counters are collected with --kernel-trace only (never with the sys/hip/hsa trace domains).
tag=${1:-r02}
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-accuracy"
timeout -s KILL 300 $B --steps 20 --warmup 5 --dump-convs > $o/bench.json 2> $o/bench.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o r1 -- $B --steps 10 --warmup 3 > $o/stats.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/pmc_mfma -o r1 -- $B --steps 3 --warmup 1 --pipeline 1 > $o/pmc_mfma.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/pmc_fetch -o r1 -- $B --steps 3 --warmup 1 --pipeline 1 > $o/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/pmc_write -o r1 -- $B --steps 3 --warmup 1 --pipeline 1 > $o/pmc_write.log 2>&1
ls $o $o/pmc_mfma | head -30; tail -2 $o/pmc_mfma.log
