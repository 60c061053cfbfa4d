#!/bin/bash
# GPU visit r01s9: per-slot pipeline probe (configs[4] on one GPU), BLS + Merkle GPU tests
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python tools/slot_pipeline_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01s9_slot_pipeline.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r01s9_pytest.txt
timeout 600 python bench.py --workload bls --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d['aggregates_k2048']
print('K=1', d['value'], 'aggregates', a['value'], a['ms_per_step'], 'registry', a['validated_key_cache']['value'], a['validated_key_cache']['ms_per_step'])"
