R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r03_trainprof2; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs"
for w in r18 r50; do
  a=""; [ $w = r50 ] && a="--workload 3d_r50_fpn3d"
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/train_$w -o r1 -- $B --steps 10 --warmup 3 --mode train $a > $o/train_$w.log 2>&1
  f=$(ls $o/train_$w/*/r1_kernel_stats.csv $o/train_$w/r1_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $o/train_${w}_kernel_stats.csv
done
ls $o
