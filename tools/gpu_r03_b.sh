cd $GRAFT_REPO_ROOT; o=gpurun_out/r03_m; mkdir -p $o
B="timeout 300 python bench.py --no-cpu-baseline --no-accuracy --no-other-configs --steps 30 --warmup 5 --h2d 0 --workload 3d_r50_fpn3d"
run() { n=$1; shift; "$@" > $o/$n.json 2> $o/$n.err; python - $o/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[2], d['value'], d['unit'], d['ms_per_step'], 'seq', d.get('sequential_clips_per_s'), r['achieved'], r['all_conv_kernels']['tflops'])
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run b1p4 $B; run b2p3 $B --batch 2 --pipeline 3; run b2p2 $B --batch 2 --pipeline 2; run b4p2 $B --batch 4 --pipeline 2; run b4p3 $B --batch 4 --pipeline 3; run b1p4x $B
