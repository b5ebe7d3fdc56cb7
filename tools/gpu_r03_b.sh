cd $GRAFT_REPO_ROOT; o=gpurun_out/r03_l; mkdir -p $o
B="timeout 300 python bench.py --no-cpu-baseline --no-accuracy --no-other-configs --steps 40 --warmup 6 --h2d 0"
$B --keyframe-dce > $o/dce.json 2> $o/dce.err; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r03_l/dce.json').read().strip().splitlines()[-1]); print('dce', d['value'], d['ms_per_step'], d.get('sequential_clips_per_s'))
except Exception as e: print('ERR', e); print(open('gpurun_out/r03_l/dce.err').read()[-1500:])
PY
bash tools/gpu_r03.sh all r03_all2
