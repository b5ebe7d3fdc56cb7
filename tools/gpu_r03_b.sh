cd $GRAFT_REPO_ROOT; o=gpurun_out/r03_j; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py -m gpu -q -x > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log; grep -E "passed|failed|Error|assert|rel err" $o/pytest.log | tail -12
B="timeout 600 python bench.py --no-cpu-baseline --no-accuracy --no-other-configs --steps 20 --warmup 4 --mode train"
for w in 3d_r18_fpn3d 3d_r50_fpn3d; do
  $B --workload $w > $o/train_$w.json 2> $o/train_$w.err
  python - $o/train_$w.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], 'ms/iter')
    except Exception as e: print(f, 'ERR', e)
PY
done
