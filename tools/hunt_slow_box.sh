#!/bin/bash
# One visit of the slow-box hunt: the instruction-fetch self-check; on a slow box also the stage probe with each build of
# the pairing kernels (and what the automatic choice picks) and the field probe.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/hunt_last.txt
import ctypes, sys, os
sys.path.insert(0, os.getcwd())
from ethereum_consensus_amd import _lib
L = _lib.load(build_if_missing=False)
assert L.ecgpu_init(0) == 0
sw = (ctypes.c_double * 4)()
for rep in range(2):
    L.ecgpu_selfcheck_ifetch_sweep(sw)
    print("sweep 8KB/64KB/256KB/1MB ms:", [round(x, 2) for x in sw], "slowdown", round(sw[3] / sw[0], 2), flush=True)
print("automatic choice of the pairing kernels:", L.ecgpu_bls_tower(), "(1 = sums of products, 2 = compact-code tower)")
open("gpurun_out/hunt_slow", "w").write("1" if sw[3] / sw[0] > 1.4 else "0")
PY
if [ "$(cat gpurun_out/hunt_slow)" = "1" ]; then
  echo "SLOW BOX"
  {
  for t in sums calls; do echo "== ECGPU_TOWER=$t"; ECGPU_TOWER=$t timeout 300 python tools/bls_probe.py 65536 2>&1 | grep "verify iter"; done
  echo "== automatic"; timeout 300 python tools/bls_probe.py 65536 2>&1 | grep "verify iter"
  } | tee gpurun_out/hunt_probe_slow.txt
  timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/hunt_bench_slow.json
fi
