#!/bin/bash
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bls.py -x -q -m gpu > gpurun_out/r01c_pytest_bls.log 2>&1
tail -5 gpurun_out/r01c_pytest_bls.log
python tools/bls_probe.py 65536 2>&1 | tee gpurun_out/r01c_probe.txt
ECGPU_LANE_PAIRING=1 python tools/bls_probe.py 65536 2>&1 | tee gpurun_out/r01c_probe_lane.txt
