#!/bin/bash
# SQ counters of the stage kernels after the register-resident accumulators (pow_x base in memory, inlined doublings)
cd /root/repo; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_FLAT"
timeout 600 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d gpurun_out/pmc_r02v_1 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aggregates --workload bls > gpurun_out/r02v_pmc_1.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r02v_1 gpurun_out/r02v_pmc_1.txt
python - <<'PY'
import re, collections
d = collections.defaultdict(dict)
for l in open("gpurun_out/r02v_pmc_1.txt"):
    m = re.match(r"(\S+?)\(.*?(SQ_[A-Z_]+)\s+dispatches\s+(\d+)\s+mean\s+([\d.]+)", l)
    if m and m.group(1).split("::")[-1] in ("k_pairing", "k_h2c", "k_sig", "k_pk_validate_w1"):
        d[m.group(1).split("::")[-1]][m.group(2)] = float(m.group(4))
for k, v in d.items():
    w, wc = v["SQ_WAVES"], v["SQ_WAVE_CYCLES"]
    print(f"{k:20s} VALU/wave {v['SQ_INSTS_VALU']/w/1e6:8.3f} M  quad/wave {wc/w/1e6:8.3f} M  active {v['SQ_ACTIVE_INST_ANY']/wc:6.1%} parked {v['SQ_WAIT_ANY']/wc:6.1%} stall {v['SQ_WAIT_INST_ANY']/wc:6.1%} scratch instr/wave {v['SQ_INSTS_FLAT']/w/1e3:7.1f} k")
PY
