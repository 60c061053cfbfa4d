#!/bin/bash
# second GPU visit of round 1: full GPU parity suite, the two-workload bench line, kernel trace, and
# the occupancy experiment for the BLS kernels (ECG_BLS_WAVES variants)
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r01b_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/r01b_pytest_gpu.log
tail -5 gpurun_out/r01b_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r01b_bench.json 2> gpurun_out/r01b_bench.err
tail -c 3000 gpurun_out/r01b_bench.json; tail -5 gpurun_out/r01b_bench.err
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01b -o r01b -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r01b_prof.log 2>&1
find gpurun_out/prof_r01b -name "*.db" | head -3
DB=$(find gpurun_out/prof_r01b -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" gpurun_out/r01b_kernel_stats.txt && head -20 gpurun_out/r01b_kernel_stats.txt
python tools/bls_probe.py 65536 262144 2>&1 | tee gpurun_out/r01b_probe_w4.txt
ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_w2.so python tools/bls_probe.py 65536 262144 2>&1 | tee gpurun_out/r01b_probe_w2.txt
ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_w1.so python tools/bls_probe.py 65536 2>&1 | tee gpurun_out/r01b_probe_w1.txt
