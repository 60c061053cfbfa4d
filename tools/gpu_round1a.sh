#!/bin/bash
# first GPU visit: Merkle parity, integer-rate probe, first bench line, kernel trace
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rocm-smi --showproductname 2>&1 | head -5 > gpurun_out/smi.log
./tools/microbench > gpurun_out/microbench.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_merkle.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_merkle -o merkle -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_merkle.log 2>&1
ls -R gpurun_out/prof_merkle | head -30
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/microbench.log; tail -3 gpurun_out/bench_merkle.log
