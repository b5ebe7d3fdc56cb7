#!/bin/bash
# call F: bisect the "invalid resource handle" of the post-region eager pass; config-5 end-to-end tool
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04f; mkdir -p $o
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-accuracy"
DAT_BENCH_RESIDENT=0 timeout 300 $B --h2d 0 > $o/a_nores_noh2d.json 2> $o/a.err; echo "resident0 h2d0 rc=$?"
timeout 300 $B --h2d 0 > $o/b_res_noh2d.json 2> $o/b.err; echo "resident1 h2d0 rc=$?"
DAT_BENCH_RESIDENT=0 timeout 300 $B --h2d 1 > $o/c_nores_h2d.json 2> $o/c.err; echo "resident0 h2d1 rc=$?"
AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 timeout 400 $B --h2d 1 > $o/d_res_h2d_serial.json 2> $o/d.err; echo "resident1 h2d1 serialized rc=$?"
for f in a b c d; do echo "== $f"; grep -v "amdgpu.ids" $o/$f.err | tail -12 | cut -c1-220; done
timeout 400 python tools/bench_config5.py > $o/config5.json 2> $o/config5.err; echo "config5 rc=$?"; cat $o/config5.json; tail -5 $o/config5.err | cut -c1-300
