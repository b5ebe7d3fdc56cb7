#!/bin/bash
# round-2 GPU call H: 64-channel WD tiles (DAT_CONV_WD=2) correctness + A/B, tiled weight packing (training), roi_align lanes
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02h; mkdir -p $o
python -m pytest tests -m gpu -q -x > $o/pytest.log 2>&1; echo "pytest default rc $?" | tee -a $o/pytest.log
tail -2 $o/pytest.log
DAT_CONV_WD=2 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "conv or forward or pointwise or stem or full_size or tube" > $o/pytest_wd2.log 2>&1; echo "pytest wd2 rc $?" | tee -a $o/pytest_wd2.log
tail -2 $o/pytest_wd2.log
for rep in 1 2; do for wd in 1 2; do
  DAT_CONV_WD=$wd python tools/bench_layers.py --arch R18 --iters 10 > $o/layers_r18_wd$wd.$rep.log 2>&1
done; done
for wd in 1 2; do
  DAT_CONV_WD=$wd python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --dump-convs > $o/bench_wd$wd.json 2> $o/bench_wd$wd.err
done
python bench.py --steps 8 --warmup 3 --mode train --no-cpu-baseline > $o/train18.json 2> $o/train18.err
DAT_PACK_SIMPLE=1 python bench.py --steps 8 --warmup 3 --mode train --no-cpu-baseline > $o/train18_simplepack.json 2> $o/train18_simplepack.err
python bench.py --steps 8 --warmup 3 --mode train --arch 50 --no-cpu-baseline > $o/train50.json 2> $o/train50.err
grep -h "TOTAL\|res2_3x3" $o/layers_*.log
python - <<PY
import json
for f in ('bench_wd1','bench_wd2','train18','train18_simplepack','train50'):
    try:
        d=json.load(open('$o/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'), d['roofline']['achieved'], d['roofline']['all_conv_kernels'])
    except Exception as e: print(f,'ERR',e)
PY
