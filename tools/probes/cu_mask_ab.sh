#!/bin/bash
# A/B of CU-masked pipeline slots: the same bench call with every slot on all 256 CUs (today) and with the forwards in flight
# confined to disjoint CU subsets, so that the HBM-bound 1x1 class of one forward runs NEXT TO the power-limited 3x3x3 MFMA class of another
# instead of taking turns on the whole chip.   gpurun -- 'bash tools/probes/cu_mask_ab.sh <tag>'   -> gpurun_out/<tag>/cu_mask_*.json
tag=${1:-cumask}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
o=$R/gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
Q="--no-cpu-baseline --no-accuracy --no-other-configs --no-rocprof-check --no-roofline --h2d 0 --steps 60 --warmup 10"
run() {  # name, DAT_SLOT_CUS, extra args
    DAT_SLOT_CUS="$2" timeout -s KILL 300 python $R/bench.py $Q $3 > $o/cu_mask_$1.json 2> $o/cu_mask_$1.err
    echo "$1 [$2] $3: $(cut -c1-140 $o/cu_mask_$1.json)"
}
for wl in 3d_r18_fpn3d 3d_r50_fpn3d; do
    run ${wl}_all_p3 "" "--workload $wl"
    run ${wl}_all_p2 "" "--workload $wl --pipeline 2"
    run ${wl}_halves_p2 "0-127,128-255" "--workload $wl --pipeline 2"
    run ${wl}_all_p4 "" "--workload $wl --pipeline 4"
    run ${wl}_halves_p4 "0-127,128-255" "--workload $wl --pipeline 4"
    run ${wl}_192_64_p2 "0-191,192-255" "--workload $wl --pipeline 2"
    run ${wl}_thirds_p3 "0-85,86-170,171-255" "--workload $wl --pipeline 3"
    run ${wl}_all_again_p3 "" "--workload $wl"
done
