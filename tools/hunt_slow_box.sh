#!/bin/bash
# One visit of the slow-box hunt: the instruction-fetch self-check; on a slow box the stage probe with each build of the
# kernels and one bench.py line (which then runs on the compact build by itself).
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/hunt_last.txt
import ctypes, sys, os
sys.path.insert(0, os.getcwd())
from ethereum_consensus_amd import _lib
L = _lib.load(build_if_missing=False)
assert L.ecgpu_init(0) == 0
sw = (ctypes.c_double * 4)()
L.ecgpu_selfcheck_ifetch_sweep(sw)
print("sweep 8KB/64KB/256KB/1MB ms:", [round(x, 2) for x in sw], "slowdown", round(sw[3] / sw[0], 2), flush=True)
open("gpurun_out/hunt_slow", "w").write("1" if sw[3] / sw[0] > 1.4 else "0")
PY
if [ "$(cat gpurun_out/hunt_slow)" = "1" ] || [ -n "$HUNT_ALWAYS" ]; then
  [ "$(cat gpurun_out/hunt_slow)" = "1" ] && echo "SLOW BOX"
  # the driver's bench line as the library dispatches it on this box; then the small-batch paths with the lane-pair end of the
  # message stage forced (its hot loops -- one 43 KB doubling -- fit the instruction cache); then the dispatch parity cases
  bash tools/gpu_visit.sh ${HUNT_TAG:-r04slow2} bench:--no-cpu-baseline env:ECGPU_H2C_SPLIT_MAX=65536 bench:--workload_bls_--no-cpu-baseline_--no-aggregates \
       unenv:ECGPU_H2C_SPLIT_MAX py:h2c_small_probe.py env:ECGPU_TOWER=sums py:h2c_small_probe.py
fi
