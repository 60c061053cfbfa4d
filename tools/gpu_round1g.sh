#!/bin/bash
# GPU visit r01g: kernel timeline of the multi-stream state root
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01g -o r01g -- python bench.py --workload merkle --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r01g_prof.log 2>&1
tail -3 gpurun_out/r01g_prof.log
DB=$(find gpurun_out/prof_r01g -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_timeline.py "$DB" 110 gpurun_out/r01g_merkle_timeline.txt && cat gpurun_out/r01g_merkle_timeline.txt
