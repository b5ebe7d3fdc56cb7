#!/bin/bash
# round-2 GPU call O: where the weights-stationary kernel spends its time (ablations: 1 no patch loads, 4 no stores)
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02o; mkdir -p $o
for ab in 0 1 4 5; do
  DAT_CONV_ABLATE=$ab timeout 200 python tools/bench_layers.py --arch R18 --iters 20 --only res2_3x3 > $o/layers_ab$ab.log 2>&1
  echo "ablate $ab: $(grep res2_3x3 $o/layers_ab$ab.log)"
done
