#!/bin/bash
# GPU visit r01end: smoke(), full GPU suite, bench line, rocprofv3 kernel stats and PMC traffic passes of the final round-1 state
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/gpu_profile_round.sh r01end
