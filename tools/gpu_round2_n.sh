#!/bin/bash
# committee batches with the key stage ordered behind the side streams' fork wait; Merkle: every chain kernel with issue priority,
# the other fields of a state root from the start (ECGPU_STATE_AUX_EARLY) against from the middle of the validator pass
cd /root/repo
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 > gpurun_out/r02n_bench_$tag.json 2> gpurun_out/r02n_err_$tag.txt; }
run default X=1
run aux_early ECGPU_STATE_AUX_EARLY=1
run pk1 ECGPU_PK_WAVES=1
python bench.py --workload epoch --steps 4 --warmup 1 > gpurun_out/r02n_epoch.json 2>> gpurun_out/r02n_err_default.txt
python bench.py --workload slots > gpurun_out/r02n_slots.json 2>> gpurun_out/r02n_err_default.txt
ECGPU_STATE_AUX_EARLY=1 python bench.py --workload slots > gpurun_out/r02n_slots_aux_early.json 2>> gpurun_out/r02n_err_default.txt
python - <<'PY'
import json
for t in ("default", "aux_early", "pk1"):
    try:
        d = json.loads(open(f"gpurun_out/r02n_bench_{t}.json").read().strip().splitlines()[-1])
        print(t, "step", round(d["ms_per_step"], 2), "agg2048", round(d["aggregates_k2048"]["ms_per_step"], 2), "reg",
              round(d["aggregates_k2048"]["validated_key_cache"]["ms_per_step"], 2), "block", round(d["block"]["reference_semantics"]["block_verify_ms"], 2),
              round(d["block"]["validated_key_registry"]["block_verify_ms"], 2), "merkle", round(d["merkle"]["ms_per_step"], 4), d["merkle"]["roofline"]["avg_launch_ms"], d["check"])
    except Exception as ex:
        print(t, "failed", ex)
for t in ("epoch", "slots", "slots_aux_early"):
    e = json.loads(open(f"gpurun_out/r02n_{t}.json").read().strip().splitlines()[-1])
    print(t, e["ms_per_step"], e.get("check"), e["roofline"].get("sub_latency_ms"))
PY
timeout 600 python -m pytest tests/test_gpu_merkle.py -m gpu -x -q 2>&1 | tail -3
