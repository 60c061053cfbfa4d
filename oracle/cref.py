"""ctypes loader of oracle/_build/liboracle.so (C restatement; test infrastructure only)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        srcs = [os.path.join(_HERE, "c", f) for f in os.listdir(os.path.join(_HERE, "c")) if f.endswith(".c")]
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        L = ctypes.CDLL(path)
        L.oc_sha256.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        L.oc_merkleize_bytes.restype = ctypes.c_uint64
        L.oc_merkleize_bytes.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64,
                                         ctypes.c_void_p]
        L.oc_htr_validators.restype = ctypes.c_uint64
        L.oc_htr_validators.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        L.oc_validators_subtree_root.restype = ctypes.c_uint64
        L.oc_validators_subtree_root.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        _lib = L
    return _lib


def merkleize_bytes(data: bytes, limit_chunks: int = 0, mix_len=None):
    out = ctypes.create_string_buffer(32)
    h = lib().oc_merkleize_bytes(data, len(data), limit_chunks, 0 if mix_len is None else 1, mix_len or 0, out)
    return out.raw, h


def htr_validators(ssz121: bytes, limit: int = 1 << 40):
    out = ctypes.create_string_buffer(32)
    h = lib().oc_htr_validators(ssz121, len(ssz121) // 121, limit, out)
    return out.raw, h


def htr_validators_threads(ssz121: bytes, threads: int, limit: int = 1 << 40):
    """The same root on `threads` host threads: thread t reduces the aligned subtree of W validators starting at t W (ctypes
    releases the GIL around the C call), the sub-roots are combined by the Python restatement.  (root, hash64 count)"""
    from concurrent.futures import ThreadPoolExecutor
    from . import ssz as O
    n = len(ssz121) // 121
    w = 1
    while w * threads < n:
        w <<= 1
    n_sub = (n + w - 1) // w if n else 0
    view = memoryview(ssz121)

    def one(t):
        lo, hi = t * w, min(n, (t + 1) * w)
        out = ctypes.create_string_buffer(32)
        part = bytes(view[121 * lo:121 * hi])
        return out, lib().oc_validators_subtree_root(part, hi - lo, w, out)

    with ThreadPoolExecutor(max_workers=max(threads, 1)) as ex:
        res = list(ex.map(one, range(n_sub)))
    hashes = sum(h for _, h in res)
    root = O.merkleize_subtree_roots([o.raw for o, _ in res], w, limit)
    top = (n_sub - 1) + (limit - 1).bit_length() - (w - 1).bit_length() if n_sub else 0  # upper bound is fine for a rate
    return O.mix_in_length(root, n), hashes + max(top, 0) + 1


def sha256(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    lib().oc_sha256(data, len(data), out)
    return out.raw
