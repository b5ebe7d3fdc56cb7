#!/bin/bash
# call K: the ADVICE low-item tests + the training tests after the bookkeeping change
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04k; mkdir -p $o
timeout 1500 python -m pytest -x -q -m gpu tests/test_gpu_train.py tests/test_gpu_train_full.py "tests/test_gpu_model.py::test_several_images_per_forward_give_each_image_its_own_results" > $o/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $o/pytest.log | cut -c1-250
