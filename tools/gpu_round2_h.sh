#!/bin/bash
TAG=${1:-r02h}
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/${TAG}_gputests.txt
tail -3 gpurun_out/${TAG}_gputests.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
ECGPU_PAIRING=vm3 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload bls > gpurun_out/${TAG}_bench_vm3.json 2> gpurun_out/${TAG}_bench_vm3.err
ECGPU_TOWER=calls ECGPU_PAIRING=lane timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload bls --no-aggregates > gpurun_out/${TAG}_bench_calls.json 2> gpurun_out/${TAG}_bench_calls.err
for f in bench bench_vm3 bench_calls; do echo "== $f"; cut -c1-300 gpurun_out/${TAG}_$f.json; tail -2 gpurun_out/${TAG}_$f.err; done
