#!/bin/bash
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_w2.so timeout 300 python tools/bls_probe.py 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01zd_probe_w2.txt
