#!/bin/bash
# GPU visit r01j: lane BLS occupancy variants after the fp_mul ABI fix; HBM traffic counters (separate passes)
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python tools/bls_probe.py 65536 2>&1 | tee gpurun_out/r01j_probe_w2.txt
for v in w1 w4; do
  ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_$v.so timeout 300 python tools/bls_probe.py 65536 2>&1 | tee gpurun_out/r01j_probe_$v.txt
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_r01j_bls_$c -- python tools/bls_probe.py 65536 > gpurun_out/r01j_pmc_bls_$c.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_r01j_bls_$c gpurun_out/r01j_pmc_bls_$c.txt; cat gpurun_out/r01j_pmc_bls_$c.txt
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_r01j_merkle_$c -- python bench.py --workload merkle --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01j_pmc_merkle_$c.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_r01j_merkle_$c gpurun_out/r01j_pmc_merkle_$c.txt; head -8 gpurun_out/r01j_pmc_merkle_$c.txt
done
du -sh gpurun_out/pmc_r01j_* | tail -5
