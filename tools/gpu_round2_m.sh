#!/bin/bash
# committee batches: key stage with room for two waves per SIMD + half-register-file side stages with issue priority (default now)
# against the round's earlier configuration; Merkle chain-tail kernels with issue priority
cd /root/repo
run() { tag=$1; shift; env "$@" python bench.py --steps 10 --warmup 3 > gpurun_out/r02m_bench_$tag.json 2> gpurun_out/r02m_err_$tag.txt; }
run default X=1
run old ECGPU_PK_WAVES=1 ECGPU_G2_WAVES=1
run pk2_g2full ECGPU_PK_WAVES=2 ECGPU_G2_WAVES=1
python bench.py --workload epoch --steps 4 --warmup 1 > gpurun_out/r02m_epoch.json 2>> gpurun_out/r02m_err_default.txt
python bench.py --workload slots > gpurun_out/r02m_slots.json 2>> gpurun_out/r02m_err_default.txt
python - <<'PY'
import json
for t in ("default", "old", "pk2_g2full"):
    try:
        d = json.loads(open(f"gpurun_out/r02m_bench_{t}.json").read().strip().splitlines()[-1])
        print(t, "step", round(d["ms_per_step"], 2), "agg2048", round(d["aggregates_k2048"]["ms_per_step"], 2), "reg",
              round(d["aggregates_k2048"]["validated_key_cache"]["ms_per_step"], 2), "block", round(d["block"]["reference_semantics"]["block_verify_ms"], 2),
              round(d["block"]["validated_key_registry"]["block_verify_ms"], 2), "merkle", round(d["merkle"]["ms_per_step"], 4), d["check"])
    except Exception as ex:
        print(t, "failed", ex)
for t in ("epoch", "slots"):
    e = json.loads(open(f"gpurun_out/r02m_{t}.json").read().strip().splitlines()[-1])
    print(t, e["ms_per_step"], e.get("check"), e["roofline"].get("sub_latency_ms"))
PY
timeout 600 python -m pytest tests/test_gpu_merkle.py -m gpu -x -q 2>&1 | tail -3
timeout 800 python -m pytest tests/test_gpu_bls.py -m gpu -x -q -k "not config2_full_size" 2>&1 | tail -3
