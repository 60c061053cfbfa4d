#!/bin/bash
# GPU visit r01s11: one shared inversion for the two SSWU maps of a message
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu_round1zi.sh r01s11
timeout 900 python -m pytest tests/test_gpu_bls.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r01s11_pytest_bls.txt
