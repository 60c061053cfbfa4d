#!/bin/bash
# GPU visit r01r: why did the lane pairing launch get slower? kernel-trace durations vs HIP-event brackets
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
ECGPU_PAIRING=lane timeout 300 python tools/bls_probe.py 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01r_probe_lane.txt
ECGPU_PAIRING=lane timeout 300 python tools/bls_probe.py 65536 65536 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r01r_probe_lane.txt
ECGPU_PAIRING=lane timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01r -o r01r -- python tools/bls_probe.py 65536 > gpurun_out/r01r_prof.log 2>&1
DB=$(find gpurun_out/prof_r01r -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_timeline.py "$DB" 14
rm -rf gpurun_out/prof_r01r
rocm-smi --showclocks --showpower 2>&1 | head -30
