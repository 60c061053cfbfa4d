"""What the FIRST BLS call of a process costs, with and without ecgpu_warmup (include/ecgpu.h; VERDICT round 5, missing 6).

    python tools/first_call_probe.py            -> one JSON line: cold first call / warm-up / first call after warm-up / steady state
    python tools/first_call_probe.py --child cold|warm

Each measurement needs a fresh process (the costs are per process), so the parent starts two children.  The vector is the
reference's own (crypto/bls.rs:530-544 test_can_sign)."""
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PK = bytes.fromhex("a3843eddcff557c1d9cc39b165688a8211979cef3679ef7c79751023dce64396f9ae6b86fa7b1fa15b9041d71dde7614")
SIG = bytes.fromhex("a01e49276730e4752eef31b0570c8707de501398dac70dd144438cd1bd05fb9b9bb3e1a9ceef0a68cc08904362cafa3f1005e5b699a41847fff6f5552260468846"
                    "de5bdbf94a9aedeb29bc6cdb2c1d34922d9e9af4c0593a69ae978a90b5aba6")
MSG = b"blst is such a blast"


def child(mode: str) -> None:
    from ethereum_consensus_amd import _lib
    L = _lib.load(build_if_missing=False)
    out = {}
    t0 = time.perf_counter()
    assert L.ecgpu_init(0) == 0
    out["init_ms"] = (time.perf_counter() - t0) * 1e3
    if mode == "warm":
        t0 = time.perf_counter()
        rc = L.ecgpu_warmup(0)
        out["warmup_ms"] = (time.perf_counter() - t0) * 1e3
        assert rc == 0, rc
    pk, sig, msg = ctypes.create_string_buffer(PK, 48), ctypes.create_string_buffer(SIG, 96), ctypes.create_string_buffer(MSG, len(MSG))
    t0 = time.perf_counter()
    rc = L.ecgpu_verify(pk, msg, len(MSG), sig)
    out["first_call_ms"] = (time.perf_counter() - t0) * 1e3
    assert rc == 0, rc
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        L.ecgpu_verify(pk, msg, len(MSG), sig)
        ts.append((time.perf_counter() - t0) * 1e3)
    out["warm_call_ms"] = sorted(ts)[len(ts) // 2]
    # the other half: a header root (the zero-hash ladder, the Merkle kernels' code objects)
    hdr, root = ctypes.create_string_buffer(112), ctypes.create_string_buffer(32)
    t0 = time.perf_counter()
    L.ecgpu_htr_beacon_block_header(hdr, root)
    out["first_header_root_ms"] = (time.perf_counter() - t0) * 1e3
    print(json.dumps({k: round(v, 3) for k, v in out.items()}))


def probe() -> dict:
    res = {}
    for mode in ("cold", "warm"):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode], capture_output=True, text=True, timeout=600, cwd=ROOT)
        if p.returncode != 0:
            raise RuntimeError(p.stdout[-1000:] + p.stderr[-1000:])
        res[mode] = json.loads(p.stdout.strip().splitlines()[-1])
    return {"cold_first_call_ms": res["cold"]["first_call_ms"], "warmup_ms": res["warm"]["warmup_ms"],
            "first_call_after_warmup_ms": res["warm"]["first_call_ms"], "warm_call_ms": res["warm"]["warm_call_ms"],
            "cold_first_header_root_ms": res["cold"]["first_header_root_ms"], "first_header_root_after_warmup_ms": res["warm"]["first_header_root_ms"]}


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        print(json.dumps(probe()))
