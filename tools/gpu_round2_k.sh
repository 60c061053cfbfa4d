#!/bin/bash
# key-validation kernel with room for 1..4 waves per SIMD (ECGPU_PK_WAVES): the default bench line, the 256 x 2048 aggregates
# and the whole-epoch workload
cd /root/repo
for W in 1 2 3 4; do
  ECGPU_PK_WAVES=$W python bench.py --steps 8 --warmup 2 > gpurun_out/r02k_bench_w$W.json 2> gpurun_out/r02k_err_w$W.txt
  ECGPU_PK_WAVES=$W python bench.py --workload epoch --steps 4 --warmup 1 > gpurun_out/r02k_epoch_w$W.json 2>> gpurun_out/r02k_err_w$W.txt
done
python - <<'PY'
import json
for W in (1, 2, 3, 4):
    try:
        d = json.loads(open(f"gpurun_out/r02k_bench_w{W}.json").read().strip().splitlines()[-1])
        e = json.loads(open(f"gpurun_out/r02k_epoch_w{W}.json").read().strip().splitlines()[-1])
        print("W", W, "step", round(d["ms_per_step"], 2), "pk", round(d["roofline"]["stage_ms"]["bls_pk_validate"], 3),
              "agg2048", round(d["aggregates_k2048"]["ms_per_step"], 2), "epoch", round(e["ms_per_step"], 2), e.get("check"))
    except Exception as ex:
        print(W, "failed", ex)
PY
