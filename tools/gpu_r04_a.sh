#!/bin/bash
# round 4, call A: the new parity gates + ADVICE tests, then a short bench with the new fields
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04a
timeout 1500 python -m pytest -x -q -s -m gpu \
  "tests/test_gpu_model.py::test_pipeline_graphs_survive_workspace_growth_and_a_second_geometry" \
  "tests/test_gpu_model.py::test_pipelined_engine_writes_the_same_detections_as_the_eager_loop" \
  "tests/test_gpu_kernels.py::test_full_size_layers_spot_checked" \
  "tests/test_gpu_parity_full.py::test_fp32_four_clips_per_forward_at_the_bench_shape_match_the_oracle_clip_by_clip" \
  "tests/test_gpu_parity_full.py::test_bf16_graph_of_four_clips_gives_every_clip_the_results_of_the_eager_one_clip_forward" \
  > gpurun_out/r04a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04a/pytest.log
tail -40 gpurun_out/r04a/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err
echo "bench rc=$?"
head -c 3000 gpurun_out/r04a/bench.json
