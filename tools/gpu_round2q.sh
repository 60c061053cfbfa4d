#!/bin/bash
# GPU visit r01s19: the compact-code pairing kernels (ECGPU_TOWER=calls) next to the sums-of-products ones: parity + time
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
{
echo "== ECGPU_TOWER=sums"; ECGPU_TOWER=sums timeout 300 python tools/bls_probe.py 65536 2>&1 | grep "verify iter"
echo "== ECGPU_TOWER=calls"; ECGPU_TOWER=calls timeout 300 python tools/bls_probe.py 65536 2>&1 | grep "verify iter"
echo "== auto"; timeout 300 python tools/bls_probe.py 65536 2>&1 | grep "verify iter"
} | tee gpurun_out/r01s19_tower_variants.txt
ECGPU_TOWER=calls ECGPU_PAIRING=lane timeout 900 python -m pytest tests/test_gpu_bls.py -m gpu -x -q 2>&1 | tail -2
ECGPU_TOWER=calls timeout 900 python -m pytest tests/test_gpu_bls.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_bls.py -m gpu -x -q 2>&1 | tail -2
