#!/bin/bash
# round-2 GPU call F: training path after the flat-buffer / cached-gradient-layer refactor
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02f; mkdir -p $o
python -m pytest tests/test_gpu_train.py -m gpu -q -x > $o/pytest_train.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest_train.log
tail -3 $o/pytest_train.log
python bench.py --steps 8 --warmup 3 --mode train --no-cpu-baseline > $o/train18.json 2> $o/train18.err; echo rc $?
python bench.py --steps 8 --warmup 3 --mode train --arch 50 --no-cpu-baseline > $o/train50.json 2> $o/train50.err; echo rc $?
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
timeout -s KILL 250 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o r1 -- python $R/bench.py --steps 6 --warmup 2 --mode train --no-cpu-baseline > $o/stats.log 2>&1
cd $R
python - <<PY
import json
for f in ('train18','train50'):
    try:
        d=json.load(open('$o/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'])
    except Exception as e: print(f,'ERR',e); print(open('$o/%s.err'%f).read()[-1500:])
PY
