#!/bin/bash
# call J: same-box A/Bs -- RoIAlign-backward sample folding; big-tile kernel in training and per layer
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04j; mkdir -p $o
T="python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline"
for rep in 1 2; do
  DAT_ROI_BWD_FOLD=0 timeout 300 $T > $o/t_nofold_$rep.json 2>/dev/null; python -c "import json;d=json.load(open('$o/t_nofold_$rep.json'));print('train r18 nofold',d['ms_per_step'])"
  timeout 300 $T > $o/t_fold_$rep.json 2>/dev/null; python -c "import json;d=json.load(open('$o/t_fold_$rep.json'));print('train r18 fold  ',d['ms_per_step'])"
  DAT_CONV_BT=1 timeout 300 $T > $o/t_bt_$rep.json 2>/dev/null; python -c "import json;d=json.load(open('$o/t_bt_$rep.json'));print('train r18 fold+bt',d['ms_per_step'])"
done
DAT_CONV_BT=1 timeout 300 $T --workload 3d_r50_fpn3d > $o/t50_bt.json 2>/dev/null; python -c "import json;d=json.load(open('$o/t50_bt.json'));print('train r50 bt',d['ms_per_step'])"
timeout 300 $T --workload 3d_r50_fpn3d > $o/t50.json 2>/dev/null; python -c "import json;d=json.load(open('$o/t50.json'));print('train r50   ',d['ms_per_step'])"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-accuracy --h2d 0 --dump-convs"
DAT_CONV_BT=1 timeout 300 $B > $o/bt_layers.json 2> $o/bt_layers.err; grep "ms/step" $o/bt_layers.err | head -24
DAT_CONV_BT=1 timeout 300 $B --workload 3d_r50_fpn3d > $o/bt50.json 2>/dev/null; python -c "import json;d=json.load(open('$o/bt50.json'));print('r50 infer bt',d['value'])"
timeout 300 $B --workload 3d_r50_fpn3d > $o/b50.json 2>/dev/null; python -c "import json;d=json.load(open('$o/b50.json'));print('r50 infer   ',d['value'])"
