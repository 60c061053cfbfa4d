#!/bin/bash
# GPU visit r01s: validated-key registry parity + aggregates line
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bls.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python bench.py --workload bls --no-cpu-baseline --steps 8 > gpurun_out/r01s_bench.json 2> gpurun_out/r01s_bench.err
tail -3 gpurun_out/r01s_bench.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r01s_bench.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["stage_ms"])
a = j["aggregates_k2048"]
print(a["value"], a["ms_per_step"], a["check"], a["validated_key_cache"])
PY
