#!/bin/bash
# end-of-round pass: whole GPU suite, smoke, the bench line, the product tool from host frames
tag=${1:-r03_final}
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/$tag; mkdir -p $o
bash $R/tools/gpu_r03.sh all $tag
cd $R
T="python tools/test_net.py --cfg configs/test_r18_fpn3d_synthetic.yaml --synthetic 128 --synthetic-weights OUTPUT_DIR /tmp/out HIP.FRAME_TRUNK_CACHE 0"
timeout 300 $T HIP.IMS_PER_FORWARD 4 HIP.PIPELINE_DEPTH 3 > $o/test_net_tool_b4_p3.log 2>&1; grep steady $o/test_net_tool_b4_p3.log | cut -c1-200
timeout 300 $T HIP.IMS_PER_FORWARD 1 HIP.PIPELINE_DEPTH 4 > $o/test_net_tool_b1_p4.log 2>&1; grep steady $o/test_net_tool_b1_p4.log | cut -c1-200
