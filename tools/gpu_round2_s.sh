#!/bin/bash
# instruction mix and stall composition of the stage kernels of the 65 536-tuple batch (SQ counters, two passes of <= 8)
cd /root/repo; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > gpurun_out/r02s_sq_counters.txt
wc -l gpurun_out/r02s_sq_counters.txt
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM"
i=1
for P in "$P1" "$P2"; do
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/pmc_r02s_$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aggregates --workload bls > gpurun_out/r02s_pmc_$i.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_r02s_$i gpurun_out/r02s_pmc_$i.txt; grep -E "k_pairing|k_h2c|k_sig|k_pk_validate" gpurun_out/r02s_pmc_$i.txt | head -40
  i=$((i+1))
done
tail -3 gpurun_out/r02s_pmc_2.log
