#!/bin/bash
# round-2 GPU call P: ws64 tile shape A/B + parity re-check + full kernel tests
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
o=$R/gpurun_out/r02p; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $o/pytest_kernels.log 2>&1; echo "pytest kernels rc $?" | tee -a $o/pytest_kernels.log
tail -3 $o/pytest_kernels.log
for v in 1 2 3; do for ab in 0 5; do
  echo "ws64=$v ablate=$ab: $(DAT_CONV_WS64=$v DAT_CONV_ABLATE=$ab timeout 200 python tools/bench_layers.py --arch R18 --iters 20 --only res2_3x3 2>&1 | grep res2_3x3)"
done; done
