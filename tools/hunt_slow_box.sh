#!/bin/bash
# One visit of the slow-box hunt: the instruction-fetch self-check; on a slow box also the field probe (loops of
# 12 KB .. 280 KB of real tower code) and the stage probe.
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
open("gpurun_out/hunt_slow", "w").write("1" if sw[3] / sw[0] > 1.4 else "0")
PY
if [ "$(cat gpurun_out/hunt_slow)" = "1" ]; then
  echo "SLOW BOX"
  timeout 300 ./tools/fpbench 2>&1 | grep -v amdgpu.ids | tee gpurun_out/hunt_fpbench_slow.txt | grep -E "fp_mul \(call\)|fp2_mul 2 sums|fp6_mul|fp12|G2 doubling|miller|hash64 chain" 
  timeout 300 python tools/bls_probe.py 65536 2>&1 | grep "verify iter" | tee gpurun_out/hunt_probe_slow.txt
  rocm-smi --showmeminfo vram --showclocks 2>/dev/null | grep -E "VRAM|sclk|mclk|fclk" | tee gpurun_out/hunt_smi_slow.txt
fi
