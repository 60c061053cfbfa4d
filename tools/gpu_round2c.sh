#!/bin/bash
# GPU visit r01s4: message stage always on the auxiliary stream; pairing kernels in their own translation unit
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu_round1zi.sh r01s4
timeout 900 python -m pytest tests/test_gpu_bls.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r01s4_pytest_bls.txt
timeout 600 python bench.py --workload bls --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r01s4_bench_bls.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r01s4_bench_bls.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['stage_ms'], d['aggregates_k2048']['value'], d['aggregates_k2048']['validated_key_cache']['value'])
PY
