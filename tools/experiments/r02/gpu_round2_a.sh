#!/bin/bash
# round-2 GPU call A: full GPU test suite + the bench workloads (no profiler)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q -s > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02a/pytest.log
tail -5 gpurun_out/r02a/pytest.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc $?"
python bench.py --steps 5 --warmup 2 --workload 2d_r50_fpn --no-cpu-baseline > gpurun_out/r02a/bench_2d.json 2> gpurun_out/r02a/bench_2d.err; echo "bench2d rc $?"
python bench.py --steps 5 --warmup 2 --arch 50 --no-cpu-baseline > gpurun_out/r02a/bench_r50.json 2> gpurun_out/r02a/bench_r50.err; echo "bench50 rc $?"
python bench.py --steps 5 --warmup 2 --mode train --no-cpu-baseline > gpurun_out/r02a/bench_train18.json 2> gpurun_out/r02a/bench_train18.err; echo "train18 rc $?"
python bench.py --steps 5 --warmup 2 --mode train --arch 50 --no-cpu-baseline > gpurun_out/r02a/bench_train50.json 2> gpurun_out/r02a/bench_train50.err; echo "train50 rc $?"
tail -c 600 gpurun_out/r02a/bench.json
