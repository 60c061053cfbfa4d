#!/bin/bash
# division-steps inversion + two-lane message stage: the whole GPU suite, then the bench lines
cd /root/repo
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r02q_gpu_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r02q_bench.json 2> gpurun_out/r02q_err.txt
python bench.py --workload slots > gpurun_out/r02q_slots.json 2>> gpurun_out/r02q_err.txt
python bench.py --workload epoch --steps 4 --warmup 1 > gpurun_out/r02q_epoch.json 2>> gpurun_out/r02q_err.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02q_bench.json").read().strip().splitlines()[-1])
print("step", round(d["ms_per_step"], 2), d["value"], {k: round(v, 2) for k, v in d["roofline"]["stage_ms"].items()}, d["roofline"]["valu_int"]["achieved"])
print("agg2048", round(d["aggregates_k2048"]["ms_per_step"], 2), "reg", round(d["aggregates_k2048"]["validated_key_cache"]["ms_per_step"], 2),
      "block", round(d["block"]["reference_semantics"]["block_verify_ms"], 2), round(d["block"]["validated_key_registry"]["block_verify_ms"], 2),
      "merkle", round(d["merkle"]["ms_per_step"], 4), d["check"], d["box_selfcheck"]["large_code_slowdown"])
for t in ("slots", "epoch"):
    e = json.loads(open(f"gpurun_out/r02q_{t}.json").read().strip().splitlines()[-1])
    print(t, e["ms_per_step"], e["value"], e.get("check"), e["roofline"].get("sub_latency_ms"))
PY
