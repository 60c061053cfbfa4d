#!/bin/bash
# GPU visit r01k: register-resident Fp2 products inside the Fp6 routines (ECG_FP2_INLINE): timing + HBM traffic
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in inl_w2 inl_w1; do
  ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_$v.so timeout 300 python tools/bls_probe.py 65536 2>&1 | tee gpurun_out/r01k_probe_$v.txt
  ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_$v.so timeout 600 python -m pytest tests/test_gpu_bls.py -x -q -m gpu 2>&1 | tail -2
done
for c in FETCH_SIZE WRITE_SIZE; do
  ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_inl_w2.so timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_r01k_bls_$c -- python tools/bls_probe.py 65536 > gpurun_out/r01k_pmc_bls_$c.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_r01k_bls_$c gpurun_out/r01k_pmc_bls_$c.txt; head -6 gpurun_out/r01k_pmc_bls_$c.txt
done
