#!/bin/bash
# round-2 GPU call C: direct-weights conv variant: correctness (conv + model tests) and same-box A/B of the layer set
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r02c; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "conv or forward or pointwise or stem" > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log
tail -3 $o/pytest.log
for wd in 0 1 0 1; do
  DAT_CONV_WD=$wd python tools/bench_layers.py --arch R18 --iters 10 > $o/layers_r18_wd$wd.$RANDOM.log 2>&1
done
for wd in 0 1; do
  DAT_CONV_WD=$wd python tools/bench_layers.py --arch R50 --iters 10 > $o/layers_r50_wd$wd.log 2>&1
  DAT_CONV_WD=$wd python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --dump-convs > $o/bench_wd$wd.json 2> $o/bench_wd$wd.err
done
grep -h TOTAL $o/layers_*.log
python - <<PY
import json
for wd in (0,1):
    d=json.load(open('$o/bench_wd%d.json'%wd)); print(wd, d['value'], d.get('sequential_clips_per_s'), d['roofline']['achieved'], d['roofline']['all_conv_kernels'])
PY
