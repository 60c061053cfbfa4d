"""Development probe: duration of the tile stage (k_tree_tiles1) and the finishing job as a function of the tree width."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ethereum_consensus_amd import _lib  # noqa: E402

L = _lib.load(build_if_missing=False)
assert L.ecgpu_init(0) == 0
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
d_root = torch.empty(32, dtype=torch.uint8, device=dev)
for lg in (10, 12, 14, 16, 17, 18, 19):
    n = 1 << lg
    d = torch.randint(0, 256, (32 * n,), dtype=torch.uint8, device=dev)
    for tag in ("merkle_tree_tiles", "merkle_tree_job"):
        pass
    L.ecgpu_prof_enable(1)
    for _ in range(5):
        assert L.ecgpu_merkleize_dev(d.data_ptr(), 32 * n, n, 0, 0, d_root.data_ptr(), s) == 0
    torch.cuda.synchronize()
    import ctypes
    out = []
    for tag in (b"merkle_tree_tiles", b"merkle_tree_job", b"merkle_pass_chunks"):
        ms, cnt = ctypes.c_double(0), ctypes.c_uint64(0)
        L.ecgpu_prof_read(tag, ctypes.byref(ms), ctypes.byref(cnt))
        out.append(f"{tag.decode()} {1e3 * ms.value / max(cnt.value, 1):7.1f} us x{cnt.value}")
    L.ecgpu_prof_enable(0)
    t0 = time.perf_counter()
    for _ in range(20):
        L.ecgpu_merkleize_dev(d.data_ptr(), 32 * n, n, 0, 0, d_root.data_ptr(), s)
    torch.cuda.synchronize()
    print(f"2^{lg} chunks: {1e6 * (time.perf_counter() - t0) / 20:7.1f} us per root | " + " | ".join(out), flush=True)
