#!/bin/bash
# GPU visit: the sum-of-products lane groups (vm3) -- parity at configs[1] size and timing against the other pairing paths
TAG=${1:-r02f}
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
{
timeout 900 python -m pytest tests/test_gpu_bls.py -m gpu -x -q -k "config2 or status_algebra or fault_injection or whole_block" 2>&1 | tail -6
for mode in vm3 vm2 lane; do
  echo "== ECGPU_PAIRING=$mode"
  ECGPU_PAIRING=$mode timeout 300 python tools/bls_probe.py 256 2048 8192 65536 2>&1 | grep -E "verify iter 1|n="
done
} 2>&1 | tee gpurun_out/${TAG}_vm3.txt
