#!/bin/bash
# point sums of committee batches with inlined additions: parity of the aggregate paths + the committee lines + kernel timeline
cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bls.py -m gpu -x -q -k "aggregate or committee or registry or block or several or multi_scalar or config4" 2>&1 | tail -3
python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r02x_bench.json 2> gpurun_out/r02x_err.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02x_bench.json").read().strip().splitlines()[-1])
print("step", round(d["ms_per_step"], 2), "agg2048", round(d["aggregates_k2048"]["ms_per_step"], 2), "reg", round(d["aggregates_k2048"]["validated_key_cache"]["ms_per_step"], 2),
      "block", round(d["block"]["reference_semantics"]["block_verify_ms"], 2), round(d["block"]["validated_key_registry"]["block_verify_ms"], 2), d["check"], round(d["box_selfcheck"]["large_code_slowdown"], 2))
PY
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02x -o r02x -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02x_prof.log 2>&1
DB=$(find gpurun_out/prof_r02x -name "*.db" | head -1)
python tools/rocpd_window.py "$DB" k_pk_validate_w2 5000 1000 26000 | grep -v "rocclr\|elementwise" | tail -22
