#!/bin/bash
# GPU visit r01zh: why are the sums-of-products kernels slow in the library but fast in the probe?  instruction-cache counters
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_r01zh_$tag -- python tools/bls_probe.py 65536 > gpurun_out/r01zh_pmc_$tag.log 2>&1
  tail -3 gpurun_out/r01zh_pmc_$tag.log
  f=$(find gpurun_out/pmc_r01zh_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k=r["Kernel_Name"][:40]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k in agg:
        print(k, {c: "%.3g"%(v/max(cnt[(k,c)],1)) for c,v in agg[k].items()})
except Exception as e:
    print("no csv", e)
PY
done
