#!/bin/bash
# GPU test suite -> gpurun_out/<tag>/pytest.log   (usage: bash tools/gpu_tests.sh <tag> [pytest args])
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-t}; shift
mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -q -s "$@" > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/$tag/pytest.log
grep -E "passed|failed|rc " gpurun_out/$tag/pytest.log | tail -5
