#!/bin/bash
# round-3 GPU recipes, one stage per gpurun call:  bash tools/gpu_r03.sh <stage> [tag]
#   newtests   the round-3 parity additions only (full-size training parity, R-50 bottleneck gradients, fused stem through RunNet)
#   profiles   rocprofv3 --kernel-trace --stats of R-50 inference, 2D R-50-FPN, R-18 / R-50 training + per-layer conv tables
#   all        the whole -m gpu suite, smoke(), default bench line
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
stage=${1:-all}; tag=${2:-r03_$stage}; o=$R/gpurun_out/$tag; mkdir -p $o
B="python $R/bench.py --no-cpu-baseline --no-accuracy --no-other-configs"
prof() {   # prof <name> <bench args...>: kernel-trace stats of a short bench run
    n=$1; shift
    (cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/$n -o r1 -- $B --steps 10 --warmup 3 "$@" > $o/$n.log 2>&1)
    f=$(ls $o/$n/*/r1_kernel_stats.csv $o/$n/r1_kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && cp $f $o/${n}_kernel_stats.csv && head -12 $f | cut -c1-160
}
case $stage in
newtests)
    timeout 1500 python -m pytest tests/test_gpu_train_full.py tests/test_gpu_model.py -m gpu -q -s -k "bench_shape or r50_bottleneck or fused_stem" > $o/pytest.log 2>&1
    echo "pytest rc $?" | tee -a $o/pytest.log; grep -E "passed|failed|rel err|median|oracle forward|Error|assert" $o/pytest.log | tail -60 ;;
profiles)
    for w in 3d_r50_fpn3d 2d_r50_fpn; do
        timeout 300 $B --steps 20 --warmup 5 --workload $w --pipeline 1 --graph 0 --dump-convs > $o/bench_$w.json 2> $o/convs_$w.txt
        prof infer_$w --workload $w
    done
    timeout 300 $B --steps 20 --warmup 5 --pipeline 1 --graph 0 --dump-convs > $o/bench_3d_r18_fpn3d.json 2> $o/convs_3d_r18_fpn3d.txt
    prof train_r18 --mode train
    prof train_r50 --mode train --workload 3d_r50_fpn3d ;;
all)
    timeout 2400 python -m pytest tests -m gpu -q -x > $o/pytest_gpu.log 2>&1; echo "pytest -m gpu rc $?" | tee -a $o/pytest_gpu.log
    tail -3 $o/pytest_gpu.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.log 2>&1; tail -1 $o/smoke.log
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc $?"
    cut -c1-600 $o/bench_default.json ;;
esac
