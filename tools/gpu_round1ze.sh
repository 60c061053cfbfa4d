#!/bin/bash
# GPU visit r01ze: experiment -- Fp products inlined everywhere (no out-of-line fp_mul at all)
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python tools/bls_probe.py 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01ze_probe_default.txt
ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_fpinl.so timeout 300 python tools/bls_probe.py 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01ze_probe_fpinl.txt
