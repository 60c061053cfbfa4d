#!/bin/bash
# GPU visit r01s3: Merkle GPU tests incl. the sharded big lists (aligned subtrees + top-of-tree job)
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_merkle.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r01s3_pytest_merkle.txt
