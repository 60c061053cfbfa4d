#!/bin/bash
# GPU visit r01zb: doubling iteration of the Miller loop inlined into miller_loop vs the default call structure
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_mdbl.so timeout 300 python tools/bls_probe.py 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01zc_probe_mdbl.txt
ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_mdbl_cyc.so timeout 300 python tools/bls_probe.py 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01zc_probe_mdbl_cyc.txt
ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_mdbl_cyc.so timeout 600 python -m pytest tests/test_gpu_bls.py -x -q -m gpu 2>&1 | tail -2
