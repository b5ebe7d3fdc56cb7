#!/bin/bash
# round-2 GPU call E: launch-plan A/B under 3 clips in flight (forced split-K / tile size through the env knobs), then the profiling passes
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02e; mkdir -p $o
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy"
for rep in 1 2; do
  $B > $o/plan_default.$rep.json 2>/dev/null
  DAT_CONV_KSPLIT=1 $B > $o/plan_ks1.$rep.json 2>/dev/null
  DAT_CONV_KSPLIT=1 DAT_CONV_BP=256 $B > $o/plan_ks1_bp256.$rep.json 2>/dev/null
  DAT_CONV_BP=256 $B > $o/plan_bp256.$rep.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/plan_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d.get('sequential_clips_per_s'), d['roofline']['all_conv_kernels']['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
bash tools/prof_r02.sh r02e_prof
