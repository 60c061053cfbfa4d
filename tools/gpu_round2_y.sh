#!/bin/bash
# which G2 stage kernels a slow-fetch box should run: compact-code build (default there) against the sums-of-products build
cd /root/repo
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['roofline']['stage_ms'].items()}, d['roofline']['kernel'], round(d['box_selfcheck']['large_code_slowdown'],2), d['check'])"; }
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-aggregates --workload bls 2>/dev/null | show default
ECGPU_TOWER=sums ECGPU_PAIRING=vm3 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-aggregates --workload bls 2>/dev/null | show sums_g2_kernels_with_lane_groups
ECGPU_TOWER=calls ECGPU_PAIRING=lane python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-aggregates --workload bls 2>/dev/null | show compact_lane_pairing
