#!/bin/bash
# GPU visit r01e: Fp2 lane-group VM for the pairing check: parity suite, stage timings for
# lane / vm / vm2 (g = 16, 8) pairing kernels
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bls.py -x -q -m gpu > gpurun_out/r01e_pytest_bls.log 2>&1
tail -5 gpurun_out/r01e_pytest_bls.log
timeout 300 python tools/bls_probe.py 65536 2>&1 | tee gpurun_out/r01e_probe_vm2_g16.txt
ECGPU_PAIRING=lane timeout 300 python tools/bls_probe.py 65536 2>&1 | tee gpurun_out/r01e_probe_lane.txt
ECGPU_PAIRING=vm timeout 300 python tools/bls_probe.py 65536 2>&1 | tee gpurun_out/r01e_probe_vm.txt
for v in g8w60 g8w120; do
  ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_$v.so timeout 300 python tools/bls_probe.py 65536 2>&1 | tee gpurun_out/r01e_probe_vm2_$v.txt
done
