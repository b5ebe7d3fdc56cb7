#!/bin/bash
# round-3 profiling passes -> gpurun_out/$1/ (then: python tools/pmc_summary.py gpurun_out/$1 profiles/r03)
#   bench.json + conv_layers   default workload (4 clips per forward), eager sequential per-layer table
#   stats      rocprofv3 --kernel-trace --stats         bench.py default (3 forwards of 4 clips in flight)
#   pmc_mfma   --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE    (--pipeline 1)
#   pmc_fetch  --pmc FETCH_SIZE ; pmc_write --pmc WRITE_SIZE   (separate passes, --pipeline 1)
#   <workload>_kernel_stats.csv: kernel-trace stats of the other BASELINE configs (R-50 inference, 2D R-50-FPN, both trainings, tube heads)
# counters are collected with --kernel-trace only (never with the sys/hip/hsa trace domains).
tag=${1:-r03prof}
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs"
timeout -s KILL 300 $B --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err
timeout -s KILL 300 $B --steps 10 --warmup 3 --pipeline 1 --graph 0 --h2d 0 --dump-convs > $o/bench_seq.json 2> $o/conv_layers.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o r1 -- $B --steps 10 --warmup 3 > $o/stats.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/pmc_mfma -o r1 -- $B --steps 3 --warmup 1 --pipeline 1 --h2d 0 > $o/pmc_mfma.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/pmc_fetch -o r1 -- $B --steps 3 --warmup 1 --pipeline 1 --h2d 0 > $o/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/pmc_write -o r1 -- $B --steps 3 --warmup 1 --pipeline 1 --h2d 0 > $o/pmc_write.log 2>&1
for f in stats pmc_mfma pmc_fetch pmc_write; do for g in $o/$f/*/r1_*.csv; do [ -f "$g" ] && mv $g $o/$f/; done; done
prof() { n=$1; shift
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/$n -o r1 -- $B --steps 10 --warmup 3 "$@" > $o/$n.log 2>&1
    f=$(ls $o/$n/*/r1_kernel_stats.csv $o/$n/r1_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $o/${n}_kernel_stats.csv; }
prof r50_infer --workload 3d_r50_fpn3d
prof d2_r50_infer --workload 2d_r50_fpn
prof train_r18 --mode train
prof train_r50 --mode train --workload 3d_r50_fpn3d
prof tube_r18_infer --workload 3d_r18_fpn3d_tube
for w in 3d_r50_fpn3d 2d_r50_fpn; do timeout -s KILL 300 $B --steps 10 --warmup 3 --workload $w --pipeline 1 --graph 0 --h2d 0 --dump-convs > $o/bench_seq_$w.json 2> $o/conv_layers_$w.txt; done
ls $o | head -40; ls $o/stats $o/pmc_mfma | head
