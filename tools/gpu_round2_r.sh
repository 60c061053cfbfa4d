#!/bin/bash
# end-of-round timing of the lane groups at every size (default dispatch of the small sizes, forced at the large ones) and the
# bench line forced onto the two box-independent builds
cd /root/repo
{
echo "== default dispatch"
timeout 300 python tools/bls_probe.py 1 256 2048 4096 8192 16384 32768 2>&1 | grep -E "verify iter 1|n="
echo "== ECGPU_PAIRING=vm3"
ECGPU_PAIRING=vm3 timeout 300 python tools/bls_probe.py 65536 2>&1 | grep -E "verify iter 1|n="
} 2>&1 | tee gpurun_out/r02r_vm3_timing.txt
ECGPU_PAIRING=vm3 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02r_bench_forced_vm3.json 2> gpurun_out/r02r_err.txt
ECGPU_TOWER=calls ECGPU_PAIRING=lane python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02r_bench_forced_compact_build.json 2>> gpurun_out/r02r_err.txt
ECGPU_TOWER=calls python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02r_bench_as_on_a_slow_box.json 2>> gpurun_out/r02r_err.txt
python - <<'PY'
import json
for t in ("forced_vm3", "forced_compact_build", "as_on_a_slow_box"):
    d = json.loads(open(f"gpurun_out/r02r_bench_{t}.json").read().strip().splitlines()[-1])
    print(t, round(d["ms_per_step"], 2), round(d["value"]), {k: round(v, 2) for k, v in d["roofline"]["stage_ms"].items()}, d["roofline"]["kernel"], d["check"],
          "agg", round(d["aggregates_k2048"]["ms_per_step"], 2), round(d["aggregates_k2048"]["validated_key_cache"]["ms_per_step"], 2),
          "block", round(d["block"]["reference_semantics"]["block_verify_ms"], 2), round(d["block"]["validated_key_registry"]["block_verify_ms"], 2))
PY
