R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r03_tp; mkdir -p $o; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/t -o r1 -- python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs --steps 10 --warmup 3 --mode train > $o/t.log 2>&1
f=$(ls $o/t/*/r1_kernel_stats.csv $o/t/r1_kernel_stats.csv 2>/dev/null | head -1); cp $f $o/train_r18_kernel_stats.csv; head -14 $f | cut -c1-150
