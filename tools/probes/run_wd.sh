cd $GRAFT_REPO_ROOT; python tools/probes/wd_dbg.py 2>&1 | grep -v "amdgpu.ids\|^sample\|^bad"
bash tools/gpu_r03_b.sh
