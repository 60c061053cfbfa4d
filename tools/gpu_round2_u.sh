#!/bin/bash
# waves of the lane pairing kernel started in 8 phases (ECGPU_STAGGER = sleep units per phase step) against all together
cd /root/repo
for S in 0 1 2 4 8; do
  ECGPU_STAGGER=$S python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-aggregates --workload bls > gpurun_out/r02u_bench_s$S.json 2> gpurun_out/r02u_err_s$S.txt
done
python - <<'PY'
import json
for S in (0, 1, 2, 4, 8):
    try:
        d = json.loads(open(f"gpurun_out/r02u_bench_s{S}.json").read().strip().splitlines()[-1])
        print("stagger", S, "step", round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["roofline"]["stage_ms"].items()}, d["check"])
    except Exception as ex:
        print(S, "failed", ex)
PY
