#!/bin/bash
# GPU visit: A/B on ONE box -- round-1 tower (Karatsuba over reduced products, lib/variants/libecgpu_old.so) vs the
# in-tree library, plus the box-health line of the field probe (an 80 KB straight-line loop: ~48 k cycles on a good box)
TAG=${1:-r01zi}
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
{
timeout 120 ./tools/fpbench 2>&1 | grep -E "fp6_mul schoolbook|fp_mul \(call\)" | head -3
for rep in 1 2; do
  if [ -f ethereum_consensus_amd/lib/variants/libecgpu_old.so ]; then
    echo "== old (round-1 tower) library, rep $rep"
    ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_old.so timeout 300 python tools/bls_probe.py 65536 2>&1 | grep "verify iter"
  fi
  echo "== in-tree library, rep $rep"
  timeout 300 python tools/bls_probe.py 65536 2>&1 | grep "verify iter"
done
} 2>&1 | tee gpurun_out/${TAG}_ab.txt
