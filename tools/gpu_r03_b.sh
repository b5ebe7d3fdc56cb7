cd $GRAFT_REPO_ROOT; o=gpurun_out/r03_n; mkdir -p $o
B="timeout 600 python bench.py --no-cpu-baseline --no-accuracy --no-other-configs --steps 30 --warmup 5 --mode train"
t() { n=$1; shift; "$@" > $o/$n.json 2> $o/$n.err; python - $o/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['ms_per_step'], 'ms/iter')
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
t r18_on $B; t r18_off env DAT_EARLY_RPN_BWD=0 $B; t r18_on2 $B; t r18_off2 env DAT_EARLY_RPN_BWD=0 $B
t r18_nofuse env DAT_FUSE_RELU_BWD=0 $B
t r50_on $B --workload 3d_r50_fpn3d; t r50_off env DAT_EARLY_RPN_BWD=0 $B --workload 3d_r50_fpn3d
