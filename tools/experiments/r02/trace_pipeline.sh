R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r02tr; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-accuracy --steps 14 --warmup 3"
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $o/co -o r1 -- $B > $o/co.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $o/fifo -o r1 -- $B --fifo > $o/fifo.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $o/eager -o r1 -- $B --graph 0 > $o/eager.log 2>&1
ls $o/co $o/fifo
