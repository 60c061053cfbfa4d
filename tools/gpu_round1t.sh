#!/bin/bash
# GPU visit r01t: state root with batched finishing jobs; hash64 rate vs occupancy
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_merkle.py -x -q -m gpu > gpurun_out/r01t_pytest_merkle.log 2>&1
tail -5 gpurun_out/r01t_pytest_merkle.log
timeout 600 python bench.py --workload merkle --no-cpu-baseline > gpurun_out/r01t_bench_merkle.json 2> gpurun_out/r01t_bench_merkle.err
cat gpurun_out/r01t_bench_merkle.json; tail -3 gpurun_out/r01t_bench_merkle.err
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01t -o r01t -- python bench.py --workload merkle --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r01t_prof.log 2>&1
DB=$(find gpurun_out/prof_r01t -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_timeline.py "$DB" 40 gpurun_out/r01t_merkle_timeline.txt && cat gpurun_out/r01t_merkle_timeline.txt
