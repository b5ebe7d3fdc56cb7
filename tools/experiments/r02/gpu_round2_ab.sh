#!/bin/bash
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
run() { echo "$1: $(env $2 timeout 200 python tools/head_probe.py 2>&1 | grep -v amdgpu | tail -1)"; }
run default "X=1"
run bp256_1percu "DAT_CONV_BP=256 DAT_CONV_KSPLIT=1 DAT_CONV_LDS_PAD=60000"
run bp256_1percu_ks2 "DAT_CONV_BP=256 DAT_CONV_KSPLIT=2 DAT_CONV_LDS_PAD=60000"
run bp256_2percu_ks1 "DAT_CONV_BP=256 DAT_CONV_KSPLIT=1"
run bp256_2percu_ks2 "DAT_CONV_BP=256 DAT_CONV_KSPLIT=2"
run bp128_ks1 "DAT_CONV_BP=128 DAT_CONV_KSPLIT=1"
run bp128_ks2 "DAT_CONV_BP=128 DAT_CONV_KSPLIT=2"
run nolinear "DAT_CONV_LINEAR=0"
