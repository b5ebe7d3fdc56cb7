cd $GRAFT_REPO_ROOT; o=gpurun_out/r03_h; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k "train_step_gradients or trainer_steps" > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log; tail -3 $o/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-accuracy --no-other-configs --steps 30 --warmup 5 > $o/dflt.json 2> $o/dflt.err; echo "rc $?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_h/dflt.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac']); print(json.dumps(d['roofline_hbm'], indent=0)[:1500])
PY
timeout 600 python bench.py --no-cpu-baseline --no-accuracy --no-other-configs --steps 30 --warmup 5 --workload 3d_r50_fpn3d > $o/r50.json 2> $o/r50.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_h/r50.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved']); print([ (e.get('kernel'), e.get('achieved'), e.get('frac')) for e in d['roofline_hbm']])
PY
mkdir -p /tmp/out; timeout 600 python tools/test_net.py --cfg configs/test_r18_fpn3d_synthetic.yaml --synthetic 96 OUTPUT_DIR /tmp/out HIP.FRAME_TRUNK_CACHE 0 TEST.SCALES "(800,)" TEST.MAX_SIZE 1333 > $o/testnet.log 2>&1; grep -E "im_detect:|test_net" $o/testnet.log | tail -3
timeout 600 python tools/test_net.py --cfg configs/test_r18_fpn3d_synthetic.yaml --synthetic 96 OUTPUT_DIR /tmp/out HIP.FRAME_TRUNK_CACHE 0 TEST.SCALES "(800,)" TEST.MAX_SIZE 1333 HIP.IMS_PER_FORWARD 4 HIP.PIPELINE_DEPTH 3 > $o/testnet_b4.log 2>&1; grep -E "im_detect:|test_net" $o/testnet_b4.log | tail -3
timeout 600 python tools/test_net.py --cfg configs/test_r18_fpn3d_synthetic.yaml --synthetic 24 OUTPUT_DIR /tmp/out HIP.FRAME_TRUNK_CACHE 0 TEST.SCALES "(800,)" TEST.MAX_SIZE 1333 HIP.PIPELINE_DEPTH 0 > $o/testnet_eager.log 2>&1; grep -E "im_detect:|test_net" $o/testnet_eager.log | tail -3
