#!/bin/bash
# second GPU visit of round 1: BLS parity + a first look at stage timings
set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bls.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r01b_pytest_bls.txt
cat gpurun_out/r01b_pytest_bls.txt
python tools/bls_probe.py 2>&1 | tee gpurun_out/r01b_bls_probe.txt
