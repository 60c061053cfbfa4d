#!/bin/bash
# GPU visit r01za: inlining levels of the tower (callee-saved register saves are the private-segment traffic)
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python tools/bls_probe.py 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01za_probe_level0.txt
for v in inl1 inl2; do
  ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_$v.so timeout 300 python tools/bls_probe.py 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01za_probe_$v.txt
  ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_$v.so timeout 600 python -m pytest tests/test_gpu_bls.py -x -q -m gpu 2>&1 | tail -2
done
