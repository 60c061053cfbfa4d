#!/bin/bash
# GPU visit r01f: field-primitive rates vs occupancy; state root with the fields on parallel streams
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 120 tools/fpbench 2>&1 | tee gpurun_out/r01f_fpbench.txt
timeout 900 python -m pytest tests/test_gpu_merkle.py -x -q -m gpu > gpurun_out/r01f_pytest_merkle.log 2>&1
tail -5 gpurun_out/r01f_pytest_merkle.log
timeout 600 python bench.py --workload merkle --no-cpu-baseline > gpurun_out/r01f_bench_merkle.json 2> gpurun_out/r01f_bench_merkle.err
cat gpurun_out/r01f_bench_merkle.json; tail -3 gpurun_out/r01f_bench_merkle.err
