#!/bin/bash
# GPU visit r01l: fused Fp6-level routines (pre-sums and Karatsuba recombination in one pass), 1 wave/SIMD
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python tools/bls_probe.py 65536 2>&1 | tee gpurun_out/r01l_probe.txt
timeout 900 python -m pytest tests/test_gpu_bls.py -x -q -m gpu 2>&1 | tail -2
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_r01l_bls_$c -- python tools/bls_probe.py 65536 > gpurun_out/r01l_pmc_bls_$c.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_r01l_bls_$c gpurun_out/r01l_pmc_bls_$c.txt; head -6 gpurun_out/r01l_pmc_bls_$c.txt
done
