"""ctypes loader of oracle/_build/liboracle_bls.so: the C++ restatement of the BLS12-381 hot path
(oracle/c/bls12_381.cpp) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Used as the full-vector checker at sizes the Python
oracle cannot reach and as bench.py's timed CPU baseline; checked against oracle/bls12_381.py in tests/test_oracle_cbls.py."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "_build", "liboracle_bls.so")
        src = os.path.join(_HERE, "c", "bls12_381.cpp")
        if not os.path.exists(path) or os.path.getmtime(src) > os.path.getmtime(path):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        L = ctypes.CDLL(path)
        L.cbls_fast_aggregate_verify.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                                 ctypes.c_int]
        L.cbls_fast_aggregate_verify_mt.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                                    ctypes.c_int, ctypes.c_int]
        L.cbls_fav_batch_k1.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_char_p]
        L.cbls_fav_batch_k1.restype = None
        L.cbls_key_validate.argtypes = [ctypes.c_char_p]
        L.cbls_sig_check.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.cbls_hash_to_g2.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.cbls_sk_to_pk.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        L.cbls_sign.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.cbls_pairing.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
        L.cbls_aggregate_verify.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p]
        L.cbls_aggregate_sigs.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p]
        L.cbls_g1_msm.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int]
        L.cbls_g2_msm.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int]
        L.cbls_init()
        _lib = L
    return _lib


def host_threads() -> int:
    return max(1, len(os.sched_getaffinity(0)))


def fast_aggregate_verify(pks, msg: bytes, sig: bytes, eth: bool = False) -> int:
    return lib().cbls_fast_aggregate_verify(b"".join(pks), len(pks), msg, len(msg), sig, 1 if eth else 0)


def fast_aggregate_verify_long(pks48: bytes, msg: bytes, sig: bytes, eth: bool = False, threads: int = 0) -> int:
    """one call over a LONG key list (keys concatenated): the key validations on all host threads, the first failure in list order decides"""
    return lib().cbls_fast_aggregate_verify_mt(pks48, len(pks48) // 48, msg, len(msg), sig, 1 if eth else 0, threads or host_threads())


def fast_aggregate_verify_batch_k1(pks48: bytes, msgs32: bytes, sigs96: bytes, threads: int = 0) -> bytes:
    n = len(sigs96) // 96
    assert len(pks48) == 48 * n and len(msgs32) == 32 * n
    out = ctypes.create_string_buffer(max(n, 1))
    lib().cbls_fav_batch_k1(pks48, msgs32, sigs96, n, threads or host_threads(), out)
    return out.raw[:n]


def key_validate(pk: bytes) -> int:
    return lib().cbls_key_validate(pk)


def sig_check(sig: bytes):
    """(decode status, in G2 by the psi test, in G2 by [r]Q == inf)"""
    a, b = ctypes.c_int(), ctypes.c_int()
    st = lib().cbls_sig_check(sig, ctypes.byref(a), ctypes.byref(b))
    return st, bool(a.value), bool(b.value)


def hash_to_g2(msg: bytes) -> bytes:
    out = ctypes.create_string_buffer(96)
    lib().cbls_hash_to_g2(msg, len(msg), out)
    return out.raw


def sk_to_pk(sk: int) -> bytes:
    out = ctypes.create_string_buffer(48)
    lib().cbls_sk_to_pk(sk.to_bytes(32, "big"), out)
    return out.raw


def sign(sk: int, msg: bytes) -> bytes:
    out = ctypes.create_string_buffer(96)
    lib().cbls_sign(sk.to_bytes(32, "big"), msg, len(msg), out)
    return out.raw


def pairing(p48: bytes, q96: bytes):
    """e(P, Q)^3 as 12 canonical integers (c0.c0.c0, c0.c0.c1, c0.c1.c0, ...), or the decode status"""
    out = ctypes.create_string_buffer(576)
    st = lib().cbls_pairing(p48, q96, out)
    if st:
        return st
    return [int.from_bytes(out.raw[48 * i:48 * i + 48], "big") for i in range(12)]


def aggregate_verify(pks, msgs, sig: bytes) -> int:
    """crypto/bls.rs:95-112 -> status (include/ecgpu.h numbering)"""
    off = [0]
    for m in msgs:
        off.append(off[-1] + len(m))
    arr = (ctypes.c_uint64 * len(off))(*off)
    return lib().cbls_aggregate_verify(b"".join(pks), len(pks), b"".join(msgs), arr, len(msgs), sig) & 0xFF


def aggregate(sigs):
    """crypto/bls.rs:79-93 for a non-empty list -> (status, 96 bytes or None)"""
    out = ctypes.create_string_buffer(96)
    st = lib().cbls_aggregate_sigs(b"".join(sigs), len(sigs), out)
    return st, (out.raw if st == 0 else None)


def g1_msm(pks, scalars):
    out = ctypes.create_string_buffer(48)
    st = lib().cbls_g1_msm(b"".join(pks), b"".join(int(k).to_bytes(32, "big") for k in scalars), len(pks), out, host_threads())
    return st, (out.raw if st == 0 else None)


def g2_msm(sigs, scalars):
    out = ctypes.create_string_buffer(96)
    st = lib().cbls_g2_msm(b"".join(sigs), b"".join(int(k).to_bytes(32, "big") for k in scalars), len(sigs), out, host_threads())
    return st, (out.raw if st == 0 else None)
