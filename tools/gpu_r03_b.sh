cd $GRAFT_REPO_ROOT; o=gpurun_out/r03_f; mkdir -p $o
B="timeout 600 python bench.py --no-cpu-baseline --no-accuracy --no-other-configs --steps 30 --warmup 5"
run() { n=$1; shift; "$@" > $o/$n.json 2> $o/$n.err; python - $o/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[2], d['value'], d['unit'], d['ms_per_step'], 'seq', d.get('sequential_clips_per_s'), r['kernel'], r['achieved'], r['all_conv_kernels'], (d.get('host_frames') or {}).get('value_including_upload'), d.get('host_path_images'))
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run tube $B --workload 3d_r18_fpn3d_tube --dump-convs
run dflt $B
run b8p2 $B --batch 8 --pipeline 2 --h2d 0
grep -v amdgpu $o/tube.err | sort -k5 -n -r | head -30
