#!/bin/bash
# GPU visit r01s8: resident state with cached validator roots -- parity tests and the per-slot root probe
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_merkle.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r01s8_pytest_merkle.txt
timeout 600 python tools/resident_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01s8_resident_probe.txt
