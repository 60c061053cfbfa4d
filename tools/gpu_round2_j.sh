#!/bin/bash
# stages of a big K = 1 batch side by side on three streams (ECGPU_FORK_BIG) against one after the other
cd /root/repo
python bench.py --steps 10 --warmup 3 > gpurun_out/r02j_bench_default.json 2> gpurun_out/r02j_err1.txt
ECGPU_FORK_BIG=1 python bench.py --steps 10 --warmup 3 > gpurun_out/r02j_bench_fork_big.json 2> gpurun_out/r02j_err2.txt
python - <<'PY'
import json
for f in ("default", "fork_big"):
    try:
        d = json.loads(open(f"gpurun_out/r02j_bench_{f}.json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["roofline"]["stage_ms"], d["box_selfcheck"]["large_code_slowdown"])
    except Exception as e:
        print(f, "failed", e)
PY
