#!/bin/bash
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
L="--arch R18 --iters 10 --cold --topdown --only fpn_lat_P2,fpn_lat_P3,fpn_post_P2"
run() { echo "$1: $(env $2 timeout 200 python tools/bench_layers.py $L 2>&1 | grep "fpn_" | awk '{print $1, $(NF-3), $(NF-2)}' | tr '\n' ';')"; }
run default "X=1"
run nt_stores "DAT_CONV_ABLATE=32"
run default "X=1"
run nt_stores "DAT_CONV_ABLATE=32"
