#!/bin/bash
# Round-2 profile: bench line (driver's default command), per-opcode issue rates, and -- for EACH build of the pairing kernels
# (ECGPU_TOWER=sums, calls; ECGPU_PAIRING=vm3) -- rocprofv3 kernel-trace stats of the same bench command plus the PMC traffic
# passes (separate runs, FETCH_SIZE and WRITE_SIZE apart).  usage: tools/gpu_profile_round2.sh TAG
TAG=${1:-r02p}
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 120 ./tools/issue_rate > gpurun_out/${TAG}_issue_rates.txt 2>&1
cat gpurun_out/${TAG}_issue_rates.txt
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cut -c1-400 gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err
for build in sums calls vm3; do
  if [ $build = vm3 ]; then export ECGPU_PAIRING=vm3 ECGPU_TOWER=sums; else export ECGPU_PAIRING=lane ECGPU_TOWER=$build; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_$build -o ${TAG}_$build -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aggregates > gpurun_out/${TAG}_${build}_prof.log 2>&1
  DB=$(find gpurun_out/prof_${TAG}_$build -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" gpurun_out/${TAG}_${build}_bench_kernel_stats.txt && head -12 gpurun_out/${TAG}_${build}_bench_kernel_stats.txt
  rm -rf gpurun_out/prof_${TAG}_$build
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_${build}_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aggregates --workload bls > gpurun_out/${TAG}_${build}_pmc_$c.log 2>&1
    python tools/pmc_summary.py gpurun_out/pmc_${TAG}_${build}_$c gpurun_out/${TAG}_${build}_pmc_$c.txt; head -6 gpurun_out/${TAG}_${build}_pmc_$c.txt
  done
done
unset ECGPU_PAIRING ECGPU_TOWER
# Merkle PMC passes (default build)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_merkle_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload merkle > gpurun_out/${TAG}_merkle_pmc_$c.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_${TAG}_merkle_$c gpurun_out/${TAG}_merkle_pmc_$c.txt; head -6 gpurun_out/${TAG}_merkle_pmc_$c.txt
done
du -sh gpurun_out | tail -1
