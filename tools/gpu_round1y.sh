#!/bin/bash
# GPU visit r01y: kernel-trace stats of the bench command without the secondary aggregates line (so that every
# k_pairing launch in the table is a 65 536-tuple launch)
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=r01y
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG} -o ${TAG} -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aggregates > gpurun_out/${TAG}_prof.log 2>&1
tail -c 600 gpurun_out/${TAG}_prof.log
DB=$(find gpurun_out/prof_${TAG} -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" gpurun_out/${TAG}_bench_kernel_stats.txt && head -14 gpurun_out/${TAG}_bench_kernel_stats.txt
rm -rf gpurun_out/prof_${TAG}
