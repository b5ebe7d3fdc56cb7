#!/bin/bash
# call U: the product training tool end to end (shipped synthetic configs)
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04u; mkdir -p $o
for c in train_r18_fpn3d_synthetic train_r18_c4_tube_synthetic train_r18_fpn3d_tube_synthetic; do
  timeout 600 python tools/train_net.py --cfg configs/$c.yaml SOLVER.MAX_ITER 6 OUTPUT_DIR /tmp/dat_train_$c > $o/$c.log 2>&1; echo "$c rc=$?"; grep -iE "iter |Error" $o/$c.log | tail -2 | cut -c1-250
done
