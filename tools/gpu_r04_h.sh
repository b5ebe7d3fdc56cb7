#!/bin/bash
# call H: the whole -m gpu suite + smoke, as the driver runs them
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04h; mkdir -p $o
t0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu > $o/pytest.log 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - t0 ))s"; tail -6 $o/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $o/smoke.log
