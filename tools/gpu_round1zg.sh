#!/bin/bash
# GPU visit r01zg: sums-of-products tower (lazy reduction) -- BLS parity tests + stage probe
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python tools/bls_probe.py 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01zg_probe.txt
timeout 900 python -m pytest tests/test_gpu_bls.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r01zg_pytest_bls.txt
