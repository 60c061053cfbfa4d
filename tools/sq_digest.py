#!/usr/bin/env python3
"""Per-wave digest of an SQ counter pass (tools/gpu_visit.sh sq -> tools/pmc_summary.py output): VALU / scratch / LDS
instructions per wave, and the share of the wave's quad-cycles spent issuing, parked on s_waitcnt, stalled at issue."""
import collections
import re
import sys

d = collections.defaultdict(dict)
for l in open(sys.argv[1]):
    m = re.match(r"(\S+?)\(.*?(SQ_[A-Z_]+)\s+dispatches\s+(\d+)\s+mean\s+([\d.]+)", l)
    if m:
        d[m.group(1).split("::")[-1]][m.group(2)] = float(m.group(4))
for k, v in d.items():
    if not all(c in v for c in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU")) or v["SQ_WAVES"] == 0 or v["SQ_INSTS_VALU"] / v["SQ_WAVES"] < 1e5:
        continue
    w, wc = v["SQ_WAVES"], v["SQ_WAVE_CYCLES"]
    print(f"{k:22s} VALU/wave {v['SQ_INSTS_VALU'] / w / 1e6:8.3f} M  quad-cycles/wave {wc / w / 1e6:8.3f} M  active {v.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.1%} "
          f"parked {v.get('SQ_WAIT_ANY', 0) / wc:6.1%} stall {v.get('SQ_WAIT_INST_ANY', 0) / wc:6.1%}  scratch instr/wave {v.get('SQ_INSTS_FLAT', 0) / w / 1e3:7.1f} k  "
          f"LDS instr/wave {v.get('SQ_INSTS_LDS', 0) / w / 1e3:7.1f} k")
