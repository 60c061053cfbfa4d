"""Stage timing probe for the BLS pipeline (development tool; bench.py is the judged harness)."""
import ctypes
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ethereum_consensus_amd import _lib  # noqa: E402

L = _lib.load(build_if_missing=False)
assert L.ecgpu_init(0) == 0
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def S(tag, i):
    return hashlib.sha256(b"ecgpu/v1/" + tag + b"/" + i.to_bytes(4, "little")).digest()


def run(n):
    dev = torch.device("cuda:0")
    sks = b"".join((1 + int.from_bytes(S(b"sk", i), "big") % (R - 1)).to_bytes(32, "big") for i in range(n))
    msgs = b"".join(S(b"msg", i) for i in range(n))
    d_sk = torch.frombuffer(bytearray(sks), dtype=torch.uint8).to(dev)
    d_msg = torch.frombuffer(bytearray(msgs), dtype=torch.uint8).to(dev)
    d_pk = torch.empty(48 * n, dtype=torch.uint8, device=dev)
    d_sig = torch.empty(96 * n, dtype=torch.uint8, device=dev)
    d_st = torch.empty(n, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    t0 = time.time()
    assert L.ecgpu_sk_to_pk_batch_dev(d_sk.data_ptr(), n, d_pk.data_ptr(), s) == 0
    torch.cuda.synchronize()
    t1 = time.time()
    assert L.ecgpu_sign_batch_dev(d_sk.data_ptr(), 32, d_msg.data_ptr(), n, d_sig.data_ptr(), s) == 0
    torch.cuda.synchronize()
    t2 = time.time()
    print(f"n={n}: keygen {1e3*(t1-t0):.1f} ms, sign {1e3*(t2-t1):.1f} ms", flush=True)
    for it in range(2):
        L.ecgpu_prof_enable(1)
        t0 = time.time()
        rc = L.ecgpu_fast_aggregate_verify_batch_dev(d_pk.data_ptr(), None, n, d_msg.data_ptr(), d_sig.data_ptr(), n, 0,
                                                     d_st.data_ptr(), s)
        assert rc == 0, (rc, L.ecgpu_last_error())
        torch.cuda.synchronize()
        dt = time.time() - t0
        bad = int((d_st != 0).sum().item())
        line = f"  verify iter {it}: {1e3*dt:.1f} ms  -> {n/dt:.0f} sigs/s, bad={bad}"
        for tag in ("bls_pk_validate", "bls_sig", "bls_h2c", "bls_pairing", "bls_vm3_a", "bls_vm3_inv", "bls_vm3_c", "bls_row_a", "bls_row_inv", "bls_row_c"):
            ms, cnt = _lib.prof_read(tag)
            line += f" | {tag} {ms:.3f}" if tag.startswith("bls_row") or n <= 1024 else f" | {tag} {ms:.1f}"
        print(line, flush=True)
        L.ecgpu_prof_enable(0)


for n in [int(a) for a in sys.argv[1:]] or (1024, 16384, 65536):
    run(n)
