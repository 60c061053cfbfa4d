#!/bin/bash
# GPU visit r01v: shuffling parity + device-side timing
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_shuffle.py -x -q -m gpu -s 2>&1 | tail -6
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r01v_shuffle_timing.txt
import ctypes, time, torch
from ethereum_consensus_amd import _lib
L = _lib.load(build_if_missing=False); assert L.ecgpu_init(0) == 0
n = 1 << 20
d_out = torch.empty(n, dtype=torch.int64, device="cuda")
seed = ctypes.create_string_buffer(bytes(range(32)), 32)
s = torch.cuda.current_stream().cuda_stream
for it in range(3):
    L.ecgpu_prof_enable(1)
    torch.cuda.synchronize(); t0 = time.time()
    assert L.ecgpu_compute_shuffled_indices_dev(None, n, seed, 90, d_out.data_ptr(), s) == 0
    torch.cuda.synchronize(); dt = time.time() - t0
    a, _ = _lib.prof_read("shuffle_sources"); b, _ = _lib.prof_read("shuffle_apply")
    print(f"compute_shuffled_indices, 2^20 indices, 90 rounds: {dt*1e3:.3f} ms wall | sources (368 640 SHA-256 blocks) {a:.3f} ms | apply {b:.3f} ms")
    L.ecgpu_prof_enable(0)
PY
