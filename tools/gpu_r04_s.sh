#!/bin/bash
# call S: the overlapped exchange over RCCL with one rank (torch.distributed and dat_allreduce_bucket)
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04s; mkdir -p $o
timeout 900 python -m pytest -x -q -s -m gpu tests/test_gpu_train.py -k "one_rank_is_the_identity or allreduce_bucket" > $o/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "one-rank|largest|passed|failed|Error" $o/pytest.log | tail -8
