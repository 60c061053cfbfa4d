#!/bin/bash
# GPU visit: vm3 timing after a generator change + the parity test of the vm3 path at 8 192 tuples
TAG=${1:-r02g}
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
{
timeout 600 python -m pytest tests/test_gpu_bls.py -m gpu -x -q -k "config2 and vm3" 2>&1 | tail -4
echo "== ECGPU_PAIRING=vm3"
ECGPU_PAIRING=vm3 timeout 300 python tools/bls_probe.py 256 2048 4096 8192 16384 32768 65536 2>&1 | grep -E "verify iter 1|n="
} 2>&1 | tee gpurun_out/${TAG}_vm3.txt
