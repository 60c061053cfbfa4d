#!/bin/bash
# Round profile: GPU suite, bench line, rocprofv3 kernel-trace stats of the same bench command, PMC traffic passes
# (separate runs, FETCH_SIZE and WRITE_SIZE apart).  usage: tools/gpu_profile_round.sh TAG
set -x
TAG=${1:-r01z}
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -4 gpurun_out/${TAG}_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 1500 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG} -o ${TAG} -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aggregates > gpurun_out/${TAG}_prof.log 2>&1
DB=$(find gpurun_out/prof_${TAG} -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" gpurun_out/${TAG}_bench_kernel_stats.txt && head -16 gpurun_out/${TAG}_bench_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aggregates > gpurun_out/${TAG}_pmc_$c.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$c gpurun_out/${TAG}_pmc_$c.txt; head -8 gpurun_out/${TAG}_pmc_$c.txt
done
rm -rf gpurun_out/prof_${TAG}/*/*.db 2>/dev/null
