#!/bin/bash
# GPU visit r01z: pairing accumulator in LDS vs the previous build (libecgpu_nounroll.so = same code before the change), same box
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_nounroll.so timeout 300 python tools/bls_probe.py 65536 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01z_probe_before.txt
timeout 300 python tools/bls_probe.py 65536 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01z_probe_lds.txt
timeout 900 python -m pytest tests/test_gpu_bls.py -x -q -m gpu 2>&1 | tail -2
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_r01z_$c -- python tools/bls_probe.py 65536 > gpurun_out/r01z_pmc_$c.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_r01z_$c gpurun_out/r01z_pmc_$c.txt; head -3 gpurun_out/r01z_pmc_$c.txt
done
