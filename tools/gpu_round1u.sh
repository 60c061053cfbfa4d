#!/bin/bash
# GPU visit r01u: single-exponentiation Fp2 root + SSWU norm decision (k_sig, k_h2c)
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python tools/bls_probe.py 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01u_probe.txt
timeout 900 python -m pytest tests/test_gpu_bls.py -x -q -m gpu 2>&1 | tail -2
