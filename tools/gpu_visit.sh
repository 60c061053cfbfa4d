#!/bin/bash
# One GPU visit, parametrised: tools/gpu_visit.sh <tag> <step> [<step> ...]   (run through gpurun from the repo root)
#   tests            whole `pytest -m gpu` suite                     -> gpurun_out/<tag>_gpu_tests.txt
#   tests:<expr>     pytest -m gpu -k <expr>
#   bench            the driver's command (python bench.py)          -> gpurun_out/<tag>_bench.json
#   bench:<args>     python bench.py <args> (underscores for spaces) -> gpurun_out/<tag>_bench_<args>.json
#   probe[:sizes]    tools/bls_probe.py stage timing                 -> gpurun_out/<tag>_probe.txt
#   stats            rocprofv3 --kernel-trace --stats of the bench   -> gpurun_out/<tag>_kernel_stats.txt
#   sq               SQ counters per wave of the stage kernels       -> gpurun_out/<tag>_sq_counters.txt
#   pmc              FETCH_SIZE / WRITE_SIZE passes of the bench     -> gpurun_out/<tag>_pmc_{fetch,write}.txt
#   lib:<name>       following steps use lib/libecgpu_<name>.so (tools/build_variant.sh); lib: alone switches back
#   env:<K=V>        export K=V for the following steps (unenv:<K> removes it)
#   to:<seconds>     timeout of the following probe steps (tt:<seconds>: of the test steps)
#   run:<binary>     a probe binary (commas for spaces)              -> gpurun_out/<tag>_<binary>.txt
#   py:<script>      python tools/<script> (commas for spaces)                  -> gpurun_out/<tag>_<script>.txt
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out
tag=$1; shift
sfx=""; TO=900; TT=1200
for step in "$@"; do
  case "$step" in
    to:*) TO=${step#to:};;
    tt:*) TT=${step#tt:};;
    unenv:*) unset "${step#unenv:}"; sfx="";;
    lib:*) n=${step#lib:}; if [ -z "$n" ]; then unset ECGPU_LIB; sfx=""; else export ECGPU_LIB=$PWD/ethereum_consensus_amd/lib/libecgpu_$n.so; sfx="_$n"; fi;;
    env:*) export "${step#env:}"; sfx="${sfx}_$(echo ${step#env:} | tr -c 'A-Za-z0-9=' '_')";;
    tests) timeout $TT python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/${tag}${sfx}_gpu_tests.txt;;
    tests:*) timeout $TT python -m pytest tests -m gpu -x -q -k "${step#tests:}" 2>&1 | tail -6 | tee gpurun_out/${tag}${sfx}_gpu_tests_k.txt;;
    bench) timeout 900 python bench.py > gpurun_out/${tag}${sfx}_bench.json 2> gpurun_out/${tag}${sfx}_bench_err.txt; tail -c 600 gpurun_out/${tag}${sfx}_bench_err.txt
           cp gpurun_out/bench_full.json gpurun_out/${tag}${sfx}_bench_full.json 2>/dev/null; python tools/bench_digest.py gpurun_out/${tag}${sfx}_bench.json;;
    bench:*) a=${step#bench:}; timeout 900 python bench.py ${a//_/ } > gpurun_out/${tag}${sfx}_bench_${a//[^A-Za-z0-9]/}.json 2> gpurun_out/${tag}${sfx}_bench_err.txt
           tail -c 600 gpurun_out/${tag}${sfx}_bench_err.txt; python tools/bench_digest.py gpurun_out/${tag}${sfx}_bench_${a//[^A-Za-z0-9]/}.json;;
    probe) timeout $TO python tools/bls_probe.py 65536 2>&1 | tee gpurun_out/${tag}${sfx}_probe.txt;;
    probe:*) s=${step#probe:}; timeout $TO python tools/bls_probe.py ${s//,/ } 2>&1 | tee gpurun_out/${tag}${sfx}_probe.txt;;
    stats) timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py --no-cpu-baseline > gpurun_out/${tag}${sfx}_stats.log 2>&1
           DB=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
           [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" gpurun_out/${tag}${sfx}_kernel_stats.txt && head -14 gpurun_out/${tag}${sfx}_kernel_stats.txt | cut -c1-70,100-200
           rm -rf gpurun_out/prof_$tag;;
    pmc) for c in FETCH_SIZE WRITE_SIZE; do
           timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aggregates --no-calibration --workload bls > gpurun_out/${tag}${sfx}_pmc_$c.log 2>&1
           python tools/pmc_summary.py gpurun_out/pmc_${tag}_$c gpurun_out/${tag}${sfx}_pmc_$c.txt; head -6 gpurun_out/${tag}${sfx}_pmc_$c.txt | cut -c1-160
         done
         # profiles/pmc_traffic.json: per-step traffic of the dominant kernels, keyed by the source hash of the kernels that ran (4 steps: 3 + 1 warm-up)
         python tools/pmc_to_json.py $tag 4 > gpurun_out/${tag}${sfx}_pmc_traffic.json && cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
         for c in FETCH_SIZE WRITE_SIZE; do rm -rf gpurun_out/pmc_${tag}_$c; done;;
    sq) P1="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_FLAT SQ_INSTS_LDS"
        timeout 900 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aggregates --no-calibration --workload bls > gpurun_out/${tag}${sfx}_sq.log 2>&1
        python tools/pmc_summary.py gpurun_out/pmc_${tag}_sq gpurun_out/${tag}${sfx}_sq_raw.txt
        python tools/sq_digest.py gpurun_out/${tag}${sfx}_sq_raw.txt | tee gpurun_out/${tag}${sfx}_sq_counters.txt
        rm -rf gpurun_out/pmc_${tag}_sq;;
    pmc_merkle) for c in FETCH_SIZE WRITE_SIZE; do
           timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-calibration --workload merkle > gpurun_out/${tag}${sfx}_merkle_pmc_$c.log 2>&1
           python tools/pmc_summary.py gpurun_out/pmc_${tag}_$c gpurun_out/${tag}${sfx}_merkle_pmc_$c.txt; head -6 gpurun_out/${tag}${sfx}_merkle_pmc_$c.txt | cut -c1-160
         done
         python tools/pmc_to_json.py $tag 4 merkle > gpurun_out/${tag}${sfx}_merkle_pmc_traffic.json && cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
         for c in FETCH_SIZE WRITE_SIZE; do rm -rf gpurun_out/pmc_${tag}_$c; done;;
    run:*) c=${step#run:}; timeout $TO ${c//,/ } 2>&1 | tee gpurun_out/${tag}${sfx}_$(basename ${c%%,*}).txt | tail -40;;
    py:*) s=${step#py:}; timeout 900 python tools/${s//,/ } 2>&1 | tee gpurun_out/${tag}${sfx}_$(echo $s | tr -c 'A-Za-z0-9' '_').txt | tail -40;;
    *) echo "unknown step $step";;
  esac
done
