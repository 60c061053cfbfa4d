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


def sha256(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    lib().oc_sha256(data, len(data), out)
    return out.raw
