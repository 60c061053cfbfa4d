#!/bin/bash
# GPU visit r01s6: modular add / sub decided on the top limb (one carry chain), exact fallback out of line
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 ./tools/fpbench 2>&1 | grep -v amdgpu.ids > gpurun_out/r01s6_fpbench.txt; grep -E "fp_add|fp12|miller|G2 doubling|fp6_mul schoolbook" gpurun_out/r01s6_fpbench.txt
bash tools/gpu_round1zi.sh r01s6
timeout 900 python -m pytest tests/test_gpu_bls.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r01s6_pytest_bls.txt
