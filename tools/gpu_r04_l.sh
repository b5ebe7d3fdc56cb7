#!/bin/bash
# call L: the big-tile kernel on every eligible layer (DAT_CONV_BT=2) vs the default rule, per layer and on the headline
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04l; mkdir -p $o
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-accuracy --h2d 0 --dump-convs"
for v in 0 1 2; do
  DAT_CONV_BT=$v timeout 300 $B > $o/bt$v.json 2> $o/bt$v.err
  python -c "import json;d=json.load(open('$o/bt$v.json'));print('BT=$v',d['value'],d['ms_per_step'],d['sequential_clips_per_s'],d['roofline']['kernel'],d['roofline']['frac'],d['roofline']['all_conv_kernels']['ms_per_step'])"
done
echo "--- per layer, BT=2"; grep "ms/step" $o/bt2.err | head -26
DAT_CONV_BT=2 timeout 300 python bench.py --workload 3d_r50_fpn3d --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-accuracy --h2d 0 > $o/r50_bt2.json 2>/dev/null; python -c "import json;d=json.load(open('$o/r50_bt2.json'));print('r50 BT=2',d['value'])"
