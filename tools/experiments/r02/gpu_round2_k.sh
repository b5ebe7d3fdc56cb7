#!/bin/bash
# round-2 GPU call K: B-fragment prefetch across taps: correctness + same-box A/B against the previous build (libdat_hip_prev.so)
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02k; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "conv or forward or pointwise or full_size" > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log
tail -2 $o/pytest.log
for rep in 1 2; do
  DAT_LIB=$R/detectandtrack_amd/libdat_hip_prev.so python tools/bench_layers.py --arch R18 --iters 10 > $o/layers_prev.$rep.log 2>&1
  python tools/bench_layers.py --arch R18 --iters 10 > $o/layers_new.$rep.log 2>&1
done
B="timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy"
DAT_LIB=$R/detectandtrack_amd/libdat_hip_prev.so $B > $o/bench_prev.json 2> $o/bench_prev.err
$B > $o/bench_new.json 2> $o/bench_new.err
DAT_LIB=$R/detectandtrack_amd/libdat_hip_prev.so $B --pipeline 1 > $o/bench_prev_p1.json 2> $o/bench_prev_p1.err
$B --pipeline 1 > $o/bench_new_p1.json 2> $o/bench_new_p1.err
grep -h "TOTAL\|fpn_post_P2\|res4_3x3x3\|res4_0_2a" $o/layers_*.log
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'), d['roofline']['achieved'], d['roofline']['all_conv_kernels']['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
