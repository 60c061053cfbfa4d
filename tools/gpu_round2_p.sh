#!/bin/bash
# message stage on two lanes per message for batches <= 32 768 tuples: parity suite + the latency-bound lines
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_bls.py -m gpu -x -q -k "not config2_full_size" 2>&1 | tail -3
run() { tag=$1; shift; env "$@" python bench.py --steps 12 --warmup 3 > gpurun_out/r02p2_bench_$tag.json 2> gpurun_out/r02p2_err_$tag.txt; }
run default X=1
run nosplit ECGPU_H2C_SPLIT_MAX=0
python bench.py --workload slots > gpurun_out/r02p2_slots.json 2>> gpurun_out/r02p2_err_default.txt
python - <<'PY'
import json
for t in ("default", "nosplit"):
    try:
        d = json.loads(open(f"gpurun_out/r02p2_bench_{t}.json").read().strip().splitlines()[-1])
        print(t, "step", round(d["ms_per_step"], 2), "agg2048", round(d["aggregates_k2048"]["ms_per_step"], 2), "reg",
              round(d["aggregates_k2048"]["validated_key_cache"]["ms_per_step"], 2), "block", round(d["block"]["reference_semantics"]["block_verify_ms"], 2),
              round(d["block"]["validated_key_registry"]["block_verify_ms"], 2), "merkle", round(d["merkle"]["ms_per_step"], 4), d["check"])
    except Exception as ex:
        print(t, "failed", ex)
e = json.loads(open("gpurun_out/r02p2_slots.json").read().strip().splitlines()[-1])
print("slots", e["ms_per_step"], e.get("check"), e["roofline"].get("sub_latency_ms"))
PY
