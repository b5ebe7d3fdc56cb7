cd $GRAFT_REPO_ROOT; o=gpurun_out/r03_i; mkdir -p $o /tmp/out
T="timeout 600 python tools/test_net.py --cfg configs/test_r18_fpn3d_synthetic.yaml --synthetic-weights"
O="OUTPUT_DIR /tmp/out HIP.FRAME_TRUNK_CACHE 0 TEST.SCALES (800,) TEST.MAX_SIZE 1333"
$T --synthetic 128 $O > $o/testnet.log 2>&1; grep -E "test_net" $o/testnet.log | tail -1
$T --synthetic 128 $O HIP.IMS_PER_FORWARD 4 HIP.PIPELINE_DEPTH 3 > $o/testnet_b4.log 2>&1; grep -E "test_net" $o/testnet_b4.log | tail -1
$T --synthetic 128 $O HIP.PIPELINE_DEPTH 1 > $o/testnet_p1.log 2>&1; grep -E "test_net" $o/testnet_p1.log | tail -1
$T --synthetic 16 $O HIP.PIPELINE_DEPTH 0 > $o/testnet_eager.log 2>&1; grep -E "im_detect:" $o/testnet_eager.log | tail -1
tail -3 $o/testnet.log
