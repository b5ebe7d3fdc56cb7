#!/bin/bash
# call I: same-box A/B of the opt-in big-tile kernel at 4 clips per forward; training step after the RoIAlign-backward change
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04i; mkdir -p $o
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-accuracy --h2d 0"
for rep in 1 2; do
  timeout 300 $B > $o/base_$rep.json 2>/dev/null; python -c "import json;d=json.load(open('$o/base_$rep.json'));print('base   ',d['value'],d['ms_per_step'],d['roofline']['frac'])"
  DAT_CONV_BT=1 timeout 300 $B > $o/bt_$rep.json 2>/dev/null; python -c "import json;d=json.load(open('$o/bt_$rep.json'));print('bigtile',d['value'],d['ms_per_step'],d['roofline']['kernel'],d['roofline']['frac'])"
done
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_train.py -k "roi_align or two_ranks or trainer_steps or gradients_match" > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest.log
timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $o/train_r18.json 2>/dev/null; python -c "import json;d=json.load(open('$o/train_r18.json'));print('train r18',d['ms_per_step'])"
timeout 300 python bench.py --mode train --workload 3d_r50_fpn3d --steps 20 --warmup 5 --no-cpu-baseline > $o/train_r50.json 2>/dev/null; python -c "import json;d=json.load(open('$o/train_r50.json'));print('train r50',d['ms_per_step'])"
