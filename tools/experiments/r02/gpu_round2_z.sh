#!/bin/bash
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
L="--arch R18 --iters 10 --cold --only res3_0_2a_s2,res4_0_2a_s2,res5_0_2a_s2"
run() { echo "$1: $(env $2 timeout 200 python tools/bench_layers.py $L 2>&1 | grep "res" | awk '{print $1, $(NF-3), $(NF-2), $(NF-1)}' | tr '\n' ';')"; }
run default "X=1"
run bp128_dense "DAT_CONV_BP=128"
run bp128_planes "DAT_CONV_BP=128 DAT_CONV_NTAP=5"
run bp128_dense_ks1 "DAT_CONV_BP=128 DAT_CONV_KSPLIT=1"
run bp128_dense_ks2 "DAT_CONV_BP=128 DAT_CONV_KSPLIT=2"
run bp256 "DAT_CONV_BP=256"
