#!/bin/bash
# round-2 GPU call S: what bounds the HBM-bound lateral 1x1 convs (cold caches, top-down add as in the network)
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
L="--arch R18 --iters 10 --cold --topdown --only fpn_lat_P2,fpn_lat_P3,res2_3x3"
run() { echo "$1: $(env $2 timeout 200 python tools/bench_layers.py $L 2>&1 | grep "fpn_lat\|res2_3x3" | awk '{print $1, $(NF-3), $(NF-2)}' | tr '\n' ';')"; }
run default "X=1"
run no_stores "DAT_CONV_ABLATE=4"
run ntap2 "DAT_CONV_NTAP=2"
run ntap0 "DAT_CONV_NTAP=0"
run bp256 "DAT_CONV_BP=256"
run wd0 "DAT_CONV_WD=0"
run hot "X=1"
echo "hot (no flush): $(timeout 200 python tools/bench_layers.py --arch R18 --iters 10 --topdown --only fpn_lat_P2,fpn_lat_P3,res2_3x3 2>&1 | grep "fpn_lat\|res2_3x3" | awk '{print $1, $(NF-3), $(NF-2)}' | tr '\n' ';')"
