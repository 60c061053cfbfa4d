#!/bin/bash
# lane groups with 16 lanes per tuple in the Miller loops and 12 in the final exponentiation: parity at every dispatch, timing at every size
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_bls.py -m gpu -x -q -k "config2 or block or error_identity or registry" 2>&1 | tail -3
{
echo "== default dispatch"
timeout 300 python tools/bls_probe.py 1 256 4096 8192 16384 32768 2>&1 | grep -E "verify iter 1|n="
echo "== ECGPU_PAIRING=vm3"
ECGPU_PAIRING=vm3 timeout 300 python tools/bls_probe.py 65536 2>&1 | grep -E "verify iter 1|n="
} 2>&1 | tee gpurun_out/r02z_vm3_mixed_lanes_timing.txt | cut -c1-60,150-330
