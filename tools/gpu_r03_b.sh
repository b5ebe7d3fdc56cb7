cd $GRAFT_REPO_ROOT; o=gpurun_out/r03_c; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "pipelined_engine or several_images" > $o/pytest_new.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest_new.log; tail -5 $o/pytest_new.log
B="timeout 300 python bench.py --no-cpu-baseline --no-accuracy --no-other-configs --steps 40 --warmup 5"
run() { n=$1; shift; $B "$@" > $o/$n.json 2> $o/$n.err; echo "$n rc $?"; python - $o/$n.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(d['value'], d['unit'], d['ms_per_step'], 'seq', d.get('sequential_clips_per_s'), r['kernel'], r['achieved'], r['all_conv_kernels'], (d.get('host_frames') or {}).get('value_including_upload'), 'hostpath', d.get('host_path_images'))
except Exception as e: print('ERR', e)
PY
}
run b1 ; run b2 --batch 2; run d2b8 --workload 2d_r50_fpn; run d2b1 --workload 2d_r50_fpn --batch 1; run r50b1 --workload 3d_r50_fpn3d; run r50b2 --workload 3d_r50_fpn3d --batch 2
run b1_seq --pipeline 1 --graph 0 --dump-convs --h2d 0; run b2_seq --batch 2 --pipeline 1 --graph 0 --dump-convs --h2d 0
run d2b8_seq --workload 2d_r50_fpn --pipeline 1 --graph 0 --dump-convs --h2d 0
for f in $o/*.err; do echo "== $f"; grep -v "amdgpu.ids\|tag " $f | tail -n 6; done
