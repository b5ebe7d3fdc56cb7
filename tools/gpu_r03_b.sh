cd $GRAFT_REPO_ROOT; o=gpurun_out/r03_e; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "weights_in_lds or large_pointwise or conv1x1_k64" > $o/pytest_lw.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest_lw.log; grep -E "^lw |passed|failed|Error|assert" $o/pytest_lw.log | tail -30
B="timeout 300 python bench.py --no-cpu-baseline --no-accuracy --no-other-configs --steps 40 --warmup 6 --h2d 0"
run() { n=$1; shift; "$@" > $o/$n.json 2> $o/$n.err; python - $o/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[2], d['value'], d['unit'], d['ms_per_step'], 'seq', d.get('sequential_clips_per_s'), r['achieved'], r['all_conv_kernels'])
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run r50_lw $B --workload 3d_r50_fpn3d; run r50_off env DAT_CONV_PWLW=0 $B --workload 3d_r50_fpn3d
run r50b2_lw $B --workload 3d_r50_fpn3d --batch 2; run r50b2_off env DAT_CONV_PWLW=0 $B --workload 3d_r50_fpn3d --batch 2
run d2_lw $B --workload 2d_r50_fpn; run d2_off env DAT_CONV_PWLW=0 $B --workload 2d_r50_fpn
run r18_lw $B --batch 4 --pipeline 3; run r18_off env DAT_CONV_PWLW=0 $B --batch 4 --pipeline 3
run r50_seq $B --workload 3d_r50_fpn3d --pipeline 1 --graph 0 --dump-convs
run r50_seq_off env DAT_CONV_PWLW=0 $B --workload 3d_r50_fpn3d --pipeline 1 --graph 0 --dump-convs
grep -h "tag  256033" $o/r50_seq.err | head -30
