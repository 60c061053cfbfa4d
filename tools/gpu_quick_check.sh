#!/bin/bash
# quick GPU check: smoke(), full GPU suite, one bench line
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['merkle']['value'], d['merkle']['ms_per_step'], d['box_selfcheck']['large_code_slowdown'])"
