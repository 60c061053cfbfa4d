#!/bin/bash
# register-resident accumulators (pow_x base in memory, inlined doublings): whole GPU suite + the bench lines
cd /root/repo
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r02w_gpu_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r02w_bench.json 2> gpurun_out/r02w_err.txt
python bench.py --workload slots > gpurun_out/r02w_slots.json 2>> gpurun_out/r02w_err.txt
python bench.py --workload epoch --steps 4 --warmup 1 > gpurun_out/r02w_epoch.json 2>> gpurun_out/r02w_err.txt
ECGPU_TOWER=calls python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02w_bench_as_on_a_slow_box.json 2>> gpurun_out/r02w_err.txt
python - <<'PY'
import json
for f in ("bench", "bench_as_on_a_slow_box"):
    d = json.loads(open(f"gpurun_out/r02w_{f}.json").read().strip().splitlines()[-1])
    print(f, "step", round(d["ms_per_step"], 2), round(d["value"]), {k: round(v, 2) for k, v in d["roofline"]["stage_ms"].items()}, {k: round(v, 1) for k, v in d["roofline"]["valu_int"]["achieved"].items()})
    print("  agg2048", round(d["aggregates_k2048"]["ms_per_step"], 2), "reg", round(d["aggregates_k2048"]["validated_key_cache"]["ms_per_step"], 2),
          "block", round(d["block"]["reference_semantics"]["block_verify_ms"], 2), round(d["block"]["validated_key_registry"]["block_verify_ms"], 2),
          "merkle", round(d["merkle"]["ms_per_step"], 4), d["check"], round(d["box_selfcheck"]["large_code_slowdown"], 2))
for t in ("slots", "epoch"):
    e = json.loads(open(f"gpurun_out/r02w_{t}.json").read().strip().splitlines()[-1])
    print(t, round(e["ms_per_step"], 2), round(e["value"], 1), e.get("check"), e["roofline"].get("sub_latency_ms"), e.get("validated_key_registry"))
PY
