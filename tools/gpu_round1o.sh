#!/bin/bash
# GPU visit r01o: full GPU suite with the config-4 / config-5 shaped tests, the updated bench line, kernel trace
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r01o_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/r01o_pytest_gpu.log
tail -15 gpurun_out/r01o_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r01o_bench.json 2> gpurun_out/r01o_bench.err
tail -c 6000 gpurun_out/r01o_bench.json; tail -5 gpurun_out/r01o_bench.err
