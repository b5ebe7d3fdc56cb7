#!/bin/bash
# round-2 GPU call M: per-layer conv dump inside the bench (heads included) + stem microbench
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02m; mkdir -p $o
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --pipeline 1 --graph 0 --dump-convs > $o/bench_dump.json 2> $o/bench_dump.err
grep -v "^\[" $o/bench_dump.err | tail -80
