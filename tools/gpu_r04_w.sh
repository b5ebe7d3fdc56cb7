#!/bin/bash
# call W: the full-shape parity file with the tightened proposal-set gate
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04w; mkdir -p $o
timeout 1500 python -m pytest -x -q -s -m gpu tests/test_gpu_parity_full.py > $o/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "rois:|frame [0-9]|passed|failed" $o/pytest.log | tail -24
