cd $GRAFT_REPO_ROOT; python tools/probes/wd_dbg.py 2>&1 | grep -v "amdgpu.ids\|^sample\|^bad"
python tools/probes/wgrad_bench.py ablate 2>&1 | grep -E "^abl0"
