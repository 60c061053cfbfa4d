#!/bin/bash
# GPU visit r01s7: box self-check in the bench line; ABI tests
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python bench.py --no-cpu-baseline --no-aggregates --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r01s7_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r01s7_bench.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['stage_ms'], d['merkle']['value'], d['box_selfcheck'])
PY
timeout 900 python -m pytest tests/test_abi.py tests/test_gpu_bls.py -m gpu -x -q 2>&1 | tail -3
