#!/bin/bash
# GPU visit r01s5: out-of-line routines read / write their operands through private-segment (address space 5) pointers
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu_round1zi.sh r01s5
timeout 900 python -m pytest tests/test_gpu_bls.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r01s5_pytest_bls.txt
