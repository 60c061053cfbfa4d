#!/bin/bash
# GPU visit r01x: compiler-flag variants of bls.hip; registry path with the message stage on the aux stream
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in os o2 nounroll; do
  ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/variants/libecgpu_$v.so timeout 300 python tools/bls_probe.py 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01x_probe_$v.txt
done
timeout 900 python bench.py --workload bls --no-cpu-baseline --steps 8 > gpurun_out/r01x_bench.json 2> gpurun_out/r01x_bench.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r01x_bench.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["stage_ms"])
a = j["aggregates_k2048"]
print(a["value"], a["ms_per_step"], a["check"], a["validated_key_cache"])
PY
timeout 900 python -m pytest tests/test_gpu_bls.py -x -q -m gpu 2>&1 | tail -2
