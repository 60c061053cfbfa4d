#!/bin/bash
# (1) side stages launched first + key stage with room for 2 / 3 waves per SIMD: 256 x 2048 aggregates, epoch
# (2) half-register-file G2 stage kernels (ECGPU_G2_WAVES): one after the other, and side by side (ECGPU_FORK_BIG)
cd /root/repo
run() { # tag, env...
  tag=$1; shift
  env "$@" python bench.py --steps 8 --warmup 2 > gpurun_out/r02l_bench_$tag.json 2> gpurun_out/r02l_err_$tag.txt
}
run w1 ECGPU_PK_WAVES=1
run w2 ECGPU_PK_WAVES=2
run w3 ECGPU_PK_WAVES=3
run g2w2 ECGPU_G2_WAVES=1
run g2w2_fork ECGPU_G2_WAVES=1 ECGPU_FORK_BIG=1
run g2w2_fork_pk4 ECGPU_G2_WAVES=1 ECGPU_FORK_BIG=1 ECGPU_PK_WAVES=4
ECGPU_PK_WAVES=2 python bench.py --workload epoch --steps 4 --warmup 1 > gpurun_out/r02l_epoch_w2.json 2>> gpurun_out/r02l_err_w2.txt
python - <<'PY'
import json
for t in ("w1", "w2", "w3", "g2w2", "g2w2_fork", "g2w2_fork_pk4"):
    try:
        d = json.loads(open(f"gpurun_out/r02l_bench_{t}.json").read().strip().splitlines()[-1])
        print(t, "step", round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["roofline"]["stage_ms"].items()},
              "agg2048", round(d["aggregates_k2048"]["ms_per_step"], 2), "reg", round(d["aggregates_k2048"]["validated_key_cache"]["ms_per_step"], 2),
              "block", d["block"]["reference_semantics"]["block_verify_ms"], d["check"])
    except Exception as ex:
        print(t, "failed", ex)
e = json.loads(open("gpurun_out/r02l_epoch_w2.json").read().strip().splitlines()[-1])
print("epoch w2", e["ms_per_step"], e.get("check"))
PY
