#!/bin/bash
# kernel timeline of one 256 x 2048-key committee batch (which stages overlap the key stage)
cd /root/repo; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_r02o -o r02o -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r02o_prof.log 2>&1
DB=$(find gpurun_out/prof_r02o -name "*.db" | head -1)
python tools/rocpd_window.py "$DB" k_pk_validate_w2 5000 12000 14000 > gpurun_out/r02o_committee_timeline.txt
cat gpurun_out/r02o_committee_timeline.txt | tail -40
