#!/bin/bash
# GPU visit r01zf: field-primitive probe incl. sums of products with one reduction (lazy-reduction tower candidate)
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 ./tools/fpbench 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01zf_fpbench.txt
