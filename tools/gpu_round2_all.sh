#!/bin/bash
# GPU visit: the whole GPU test-suite, the bench line (both halves), the epoch and slots workloads
TAG=${1:-r02d}
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/${TAG}_gputests.txt
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --workload epoch --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_epoch.json 2> gpurun_out/${TAG}_bench_epoch.err
timeout 600 python bench.py --workload slots --steps 64 --warmup 2 > gpurun_out/${TAG}_bench_slots.json 2> gpurun_out/${TAG}_bench_slots.err
tail -3 gpurun_out/${TAG}_gputests.txt
for f in bench bench_epoch bench_slots; do echo "== $f"; cut -c1-600 gpurun_out/${TAG}_$f.json; tail -3 gpurun_out/${TAG}_$f.err; done
