#!/bin/bash
# GPU visit r01m: lazy product-operand additions; 1 vs 2 waves/SIMD again now that the traffic is down
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python tools/bls_probe.py 65536 2>&1 | tee gpurun_out/r01m_probe_w1.txt
ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_w2.so timeout 300 python tools/bls_probe.py 65536 2>&1 | tee gpurun_out/r01m_probe_w2.txt
timeout 900 python -m pytest tests/test_gpu_bls.py -x -q -m gpu 2>&1 | tail -2
