#!/bin/bash
# round-2 GPU call AC: 320-position linear tiles for grids just above one block per CU
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02ac; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_parity_full.py -m gpu -q -x -k "conv or forward or oracle" > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log
tail -3 $o/pytest.log
for n in 83 84 100 104 105 120; do echo "rois $n: lin320 $(HEAD_ROIS=$n python tools/head_probe.py conv_fcn 2>&1 | tail -1) | off $(DAT_CONV_LINEAR=3 HEAD_ROIS=$n python tools/head_probe.py conv_fcn 2>&1 | tail -1)"; done
B="timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy"
for rep in 1 2; do
DAT_CONV_LINEAR=3 $B > $o/bench_off.$rep.json 2> $o/bench_off.$rep.err
$B > $o/bench_on.$rep.json 2> $o/bench_on.$rep.err
done
DAT_CONV_LINEAR=3 $B --pipeline 1 > $o/bench_off_p1.json 2> $o/bench_off_p1.err
$B --pipeline 1 > $o/bench_on_p1.json 2> $o/bench_on_p1.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline',{})
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'), r.get('achieved'), r.get('all_conv_kernels',{}).get('ms_per_step'))
    except Exception as e: print(f,'ERR',e)
PY
