cd $GRAFT_REPO_ROOT; o=gpurun_out/r03_o; mkdir -p $o
B="timeout 600 python bench.py --no-cpu-baseline --no-accuracy --no-other-configs --steps 30 --warmup 5"
t() { n=$1; shift; "$@" > $o/$n.json 2> $o/$n.err; python - $o/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['all_conv_kernels'])
except Exception as e: print(sys.argv[2], 'ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
t train18 $B --mode train; t train50 $B --mode train --workload 3d_r50_fpn3d; t train18b $B --mode train
