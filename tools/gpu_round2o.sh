#!/bin/bash
# GPU visit r01s17: one auxiliary stream set (state roots st[0]/st[1], BLS st[2]/st[1]): state root after the aggregates, slot pipeline
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
run() { python bench.py "$@" --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); m=d.get('merkle', d); a=d.get('aggregates_k2048')
print('   merkle', m['value']/1e9, 'G leaves/s', m['ms_per_step'], 'ms', '| top-level', d['metric'], d['ms_per_step'], '| aggregates', a and (a['ms_per_step'], a['validated_key_cache']['ms_per_step']))"; }
{
echo "== merkle alone"; run --workload merkle
echo "== both (BLS + aggregates first)"; run
timeout 900 python tools/slot_pipeline_probe.py 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r01s17_one_aux_set.txt
