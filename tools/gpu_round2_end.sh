#!/bin/bash
# end of round: rocprofv3 kernel-trace stats of the driver's bench command on the box at hand (+ the plain bench line from the same box)
cd /root/repo; export TMPDIR=/tmp
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02end2_bench.json 2> gpurun_out/r02end2_err.txt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02end2 -o r02end2 -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02end2_prof.log 2>&1
DB=$(find gpurun_out/prof_r02end2 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" gpurun_out/r02end2_bench_kernel_stats.txt && head -16 gpurun_out/r02end2_bench_kernel_stats.txt | cut -c1-70,100-200
rm -rf gpurun_out/prof_r02end2
python -c "
import json
d=json.loads(open('gpurun_out/r02end2_bench.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'],2), round(d['value']), d['roofline']['kernel'], d['roofline']['avg_launch_ms'], {k:round(v,2) for k,v in d['roofline']['stage_ms'].items()}, d['box_selfcheck']['large_code_slowdown'], d['roofline']['traffic'], d['merkle']['ms_per_step'], d['merkle']['roofline']['avg_launch_ms'])"
