set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01c
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout -s KILL 250 python $R/bench.py --steps 20 --warmup 4 --dump-convs > $R/gpurun_out/r01c/bench.json 2> $R/gpurun_out/r01c/bench.err
timeout -s KILL 250 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01c/stats -o r1 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r01c/stats.log 2>&1
timeout -s KILL 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01c/pmc_fetch -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --pipeline 1 > $R/gpurun_out/r01c/pmc_fetch.log 2>&1
timeout -s KILL 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01c/pmc_write -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --pipeline 1 > $R/gpurun_out/r01c/pmc_write.log 2>&1
ls -la $R/gpurun_out/r01c/*
cat $R/gpurun_out/r01c/bench.json
