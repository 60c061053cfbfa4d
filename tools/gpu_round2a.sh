#!/bin/bash
# GPU visit r01s2: smoke(), full GPU suite, bench line and profiles of the sums-of-products tower
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 120 ./tools/fpbench 2>&1 | grep -E "fp6_mul schoolbook" | head -1
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/gpu_profile_round.sh r01s2
