#!/bin/bash
# SQ counters of the lane-group pairing kernels at 65 536 tuples (forced)
cd /root/repo; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"
i=1
for P in "$P1" "$P2"; do
  ECGPU_PAIRING=vm3 timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/pmc_r02t_$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aggregates --workload bls > gpurun_out/r02t_pmc_$i.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_r02t_$i gpurun_out/r02t_pmc_$i.txt; grep -E "k_vm3" gpurun_out/r02t_pmc_$i.txt | sed 's/.*unsigned i[a-z ]*//' | head -40
  i=$((i+1))
done
