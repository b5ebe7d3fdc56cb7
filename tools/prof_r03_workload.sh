#!/bin/bash
# PMC passes (MFMA busy, FETCH_SIZE, WRITE_SIZE; separate --pmc runs, --kernel-trace only) of another bench workload:
#   bash tools/prof_r03_workload.sh <tag> --workload 3d_r50_fpn3d     -> gpurun_out/<tag>/ ; then
#   python tools/pmc_summary.py gpurun_out/<tag> profiles/r03/<name> -
tag=$1; shift
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs --h2d 0 $*"
timeout -s KILL 300 $B --steps 10 --warmup 3 > $o/bench.json 2> $o/bench.err
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/pmc_mfma -o r1 -- $B --steps 3 --warmup 1 --pipeline 1 > $o/pmc_mfma.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/pmc_fetch -o r1 -- $B --steps 3 --warmup 1 --pipeline 1 > $o/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/pmc_write -o r1 -- $B --steps 3 --warmup 1 --pipeline 1 > $o/pmc_write.log 2>&1
for f in pmc_mfma pmc_fetch pmc_write; do for g in $o/$f/*/r1_*.csv; do [ -f "$g" ] && mv $g $o/$f/; done; done
ls $o/pmc_mfma | head -3
