cd $GRAFT_REPO_ROOT; o=gpurun_out/r03_k; mkdir -p $o
B="timeout 300 python bench.py --no-cpu-baseline --no-accuracy --no-other-configs --steps 40 --warmup 6 --h2d 0"
run() { n=$1; shift; "$@" > $o/$n.json 2> $o/$n.err; python - $o/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[2], d['value'], d['unit'], d['ms_per_step'], 'seq', d.get('sequential_clips_per_s'), r['kernel'], r['achieved'], r['all_conv_kernels']['tflops'], 'hostpath', d.get('host_path_images'))
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run base $B; run bt env DAT_CONV_BT=1 $B; run base2 $B; run bt2 env DAT_CONV_BT=1 $B
run r50 $B --workload 3d_r50_fpn3d; run r50bt env DAT_CONV_BT=1 $B --workload 3d_r50_fpn3d
run d2 $B --workload 2d_r50_fpn
