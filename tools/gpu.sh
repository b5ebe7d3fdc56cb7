#!/bin/bash
# ONE script for every GPU-box pass (replaces the per-call tools/gpu_r0N_*.sh / prof_r0N*.sh of rounds 2-4).
#   gpurun -- 'bash tools/gpu.sh <tag> <stage> [<stage> ...]'        results -> gpurun_out/<tag>/
# A stage is NAME or NAME:ARGS (ARGS: extra command-line arguments, ';' instead of spaces).  Stages:
#   tests[:pytest args]     pytest -m gpu (-> pytest_<n>.log; args are eval'ed)   e.g.  'tests:-k;"affine;or;known_answer"'
#   testsall[:pytest args]  the same without -x (every failure listed)
#   bench[:bench args]      python bench.py (-> bench.json / bench.err)        e.g.  bench:--no-other-configs
#   benchq[:bench args]     bench.py without the CPU baselines / other configs / accuracy leg (quick A/B runs; -> benchq_<n>.json)
#   layers[:bench args]     eager sequential per-layer conv table (--pipeline 1 --graph 0 --dump-convs -> conv_layers.txt)
#   stats[:bench args]      rocprofv3 --kernel-trace --stats of a --pipeline 1 run (-> <name>_kernel_stats.csv; name = stats or NAME= prefix)
#   pmc[:bench args]        three separate --pmc passes (MFMA busy / FETCH_SIZE / WRITE_SIZE; --kernel-trace only) of a --pipeline 1 run
#   trace[:bench args]      rocprofv3 --kernel-trace of a short --pipeline 1 run (-> <name>_kernel_trace.csv; tools/probes/trace_iter.py lists one iteration)
#   smoke                   __graft_entry__.smoke()
#   py:<script;args>        python <script> <args>   (-> py_<n>.log; args are eval'ed: -k;'"a;or;b"' keeps an expression together)
#   env:<NAME=V;NAME2=V2>   export for the stages that follow (A/B pairs inside one call);  unset:<NAME;NAME2>
# Environment switches (DAT_*) are inherited, so A/B pairs are two stages in one call:  DAT_X=1 bash tools/gpu.sh ...
# Allocation-poison pass of the suite (tests/conftest.py, csrc/c_api.hip):  DAT_POISON=1 DAT_WS_POISON=1 bash tools/gpu.sh <tag> tests
tag=${1:-t}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
o=$R/gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
Q="--no-cpu-baseline --no-accuracy --no-other-configs --no-rocprof-check"
n=0
for st in "$@"; do
    n=$((n + 1))
    name=${st%%:*}; args=""; [ "$st" != "$name" ] && args=$(echo "${st#*:}" | tr ';' ' ')
    out=stats; case "$args" in NAME=*) out=${args%% *}; out=${out#NAME=}; args=${args#NAME=$out}; esac
    echo "== stage $n: $name $args"
    case $name in
    tests)  (cd $R && eval "timeout -s KILL 1500 python -m pytest tests -m gpu -q -x $args" > $o/pytest_$n.log 2>&1; echo "pytest rc $?" >> $o/pytest_$n.log)
            grep -E "passed|failed|error|rc " $o/pytest_$n.log | tail -6 ;;
    testsall) (cd $R && eval "timeout -s KILL 2400 python -m pytest tests -m gpu -q $args" > $o/pytest_$n.log 2>&1; echo "pytest rc $?" >> $o/pytest_$n.log)
            grep -E "passed|failed|error|rc |^FAILED|^ERROR" $o/pytest_$n.log | tail -40 ;;
    bench)  DAT_BENCH_KEEP_ROCPROF=$o timeout -s KILL 900 python $R/bench.py $args > $o/bench.json 2> $o/bench.err; cut -c1-400 $o/bench.json ;;
    benchq) timeout -s KILL 400 python $R/bench.py $Q $args > $o/benchq_$n.json 2> $o/benchq_$n.err; cut -c1-300 $o/benchq_$n.json ;;
    layers) timeout -s KILL 400 python $R/bench.py $Q --h2d 0 --steps 10 --warmup 3 --pipeline 1 --graph 0 --dump-convs $args > $o/bench_seq_$n.json 2> $o/conv_layers_$n.txt ;;
    stats)  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/_st -o r1 -- python $R/bench.py $Q --no-roofline --h2d 0 --steps 10 --warmup 3 --pipeline 1 $args > $o/${out}.log 2>&1
            f=$(ls $o/_st/*/r1_kernel_stats.csv $o/_st/r1_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $o/${out}_kernel_stats.csv; rm -rf $o/_st
            grep -h '"metric"' $o/${out}.log | cut -c1-200; head -8 $o/${out}_kernel_stats.csv | cut -c1-160 ;;
    trace)  timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d $o/_tr -o r1 -- python $R/bench.py $Q --no-roofline --h2d 0 --steps 3 --warmup 1 --pipeline 1 $args > $o/${out}_trace.log 2>&1
            f=$(ls $o/_tr/*/r1_kernel_trace.csv $o/_tr/r1_kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $o/${out}_kernel_trace.csv; rm -rf $o/_tr
            ls -la $o/${out}_kernel_trace.csv ;;
    pmc)    B="python $R/bench.py $Q --no-roofline --h2d 0 --steps 3 --warmup 1 --pipeline 1 $args"
            timeout -s KILL 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/pmc_mfma -o r1 -- $B > $o/pmc_mfma.log 2>&1
            timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/pmc_fetch -o r1 -- $B > $o/pmc_fetch.log 2>&1
            timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/pmc_write -o r1 -- $B > $o/pmc_write.log 2>&1
            for f in pmc_mfma pmc_fetch pmc_write; do for g in $o/$f/*/r1_*.csv; do [ -f "$g" ] && mv $g $o/$f/; done; done
            ls $o/pmc_mfma | head -3 ;;
    smoke)  (cd $R && timeout -s KILL 600 python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; tail -2 $o/smoke.log) ;;
    env)    for kv in $args; do export "$kv"; done ;;
    unset)  for kv in $args; do unset "$kv"; done ;;
    py)     (cd $R && eval "timeout -s KILL 900 python $args" > $o/py_$n.log 2>&1; echo "rc $?" >> $o/py_$n.log; tail -25 $o/py_$n.log) ;;
    *)      echo "unknown stage $name" ;;
    esac
done
ls -la $o | head -40
