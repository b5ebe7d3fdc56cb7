#!/bin/bash
# call Q: clips per forward x forwards in flight, re-swept after the launch clean-up (same box)
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04q; mkdir -p $o
for cfg in "4 3" "4 2" "4 4" "6 2" "6 3" "8 2" "3 4" "4 3"; do
  set -- $cfg
  timeout 300 python bench.py --steps 30 --warmup 5 --batch $1 --pipeline $2 --no-cpu-baseline --no-other-configs --no-accuracy --h2d 0 > $o/b$1_p$2.json 2>/dev/null
  python -c "import json;d=json.load(open('$o/b$1_p$2.json'));print('batch $1 x in flight $2:',d['value'],'clips/s',d['ms_per_step'],'ms/step')"
done
