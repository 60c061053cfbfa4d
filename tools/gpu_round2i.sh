#!/bin/bash
# GPU visit r01s10: doubling formula on sums of products (G1, G2)
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 ./tools/fpbench 2>&1 | grep -v amdgpu.ids > gpurun_out/r01s10_fpbench.txt; grep -E "G2 doubling|fp6_mul schoolbook" gpurun_out/r01s10_fpbench.txt
bash tools/gpu_round1zi.sh r01s10
timeout 900 python -m pytest tests/test_gpu_bls.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r01s10_pytest_bls.txt
timeout 600 python bench.py --workload bls --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d['aggregates_k2048']
print('K=1', d['value'], d['roofline']['stage_ms'], 'aggregates', a['value'], a['ms_per_step'], 'registry', a['validated_key_cache']['value'], a['validated_key_cache']['ms_per_step'])"
