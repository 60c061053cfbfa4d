#!/bin/bash
# GPU visit r01s: pairing-check latency vs batch size, lane kernel vs lane-group VM; aggregates line with the fork
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for m in lane vm2; do
  ECGPU_PAIRING=$m timeout 600 python tools/bls_probe.py 256 2048 8192 16384 32768 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01s_probe_sizes_$m.txt
done
timeout 900 python -m pytest tests/test_gpu_bls.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python bench.py --workload bls --no-cpu-baseline --steps 5 > gpurun_out/r01s_bench.json 2> gpurun_out/r01s_bench.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r01s_bench.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["stage_ms"])
print(j["aggregates_k2048"]["value"], j["aggregates_k2048"]["ms_per_step"], j["aggregates_k2048"]["check"])
PY
