#!/bin/bash
# per-frame linear strips of the 3x3x3 layers: tests, then A/B (DAT_CONV_LINEAR=9 = off) of the per-layer table and the bench value
tag=${1:-lin_ab}
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/$tag; mkdir -p $o
cd $R
timeout -s KILL 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "conv3d or linear or block_order or big_tile" > $o/pytest.log 2>&1; tail -3 $o/pytest.log
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
B="python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs"
for lin in 9 1; do
  DAT_CONV_LINEAR=$lin timeout -s KILL 300 $B --steps 10 --warmup 3 --pipeline 1 --graph 0 --h2d 0 --dump-convs > $o/seq_$lin.json 2> $o/conv_layers_$lin.txt
done
for lin in 9 1 9 1; do DAT_CONV_LINEAR=$lin timeout -s KILL 300 $B --steps 40 --warmup 5 --h2d 0 > $o/bench_${lin}_$RANDOM.json 2>> $o/bench.err; done
for w in 3d_r50_fpn3d; do for lin in 9 1; do DAT_CONV_LINEAR=$lin timeout -s KILL 300 $B --steps 20 --warmup 5 --h2d 0 --workload $w > $o/bench_${w}_${lin}.json 2>> $o/bench.err; done; done
for lin in 9 1; do DAT_CONV_LINEAR=$lin timeout -s KILL 300 $B --steps 20 --warmup 5 --mode train > $o/train_${lin}.json 2>> $o/bench.err; done
ls $o
