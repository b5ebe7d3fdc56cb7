#!/bin/bash
# round-2 GPU call G: unrolled-tap conv variants: correctness, then same-box A/B (DAT_CONV_NTAP=0 is the table-driven WD loop)
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02g; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_parity_full.py -m gpu -q -x -k "conv or forward or pointwise or stem or full_size or bench_shape" > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log
tail -3 $o/pytest.log
for rep in 1 2; do for nt in 0 1; do
  DAT_CONV_NTAP=$nt python tools/bench_layers.py --arch R18 --iters 10 > $o/layers_r18_nt$nt.$rep.log 2>&1
done; done
for nt in 0 1; do
  DAT_CONV_NTAP=$nt python tools/bench_layers.py --arch R50 --iters 10 > $o/layers_r50_nt$nt.log 2>&1
  DAT_CONV_NTAP=$nt python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --dump-convs > $o/bench_nt$nt.json 2> $o/bench_nt$nt.err
done
grep -h TOTAL $o/layers_*.log
python - <<PY
import json
for nt in (0,1):
    d=json.load(open('$o/bench_nt%d.json'%nt)); print(nt, d['value'], d.get('sequential_clips_per_s'), d['roofline']['achieved'], d['roofline']['all_conv_kernels'])
PY
