"""ctypes loader for tests/hostsim/libhostsim.so (CPU simulator of the device lane programs)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_lib = None


def lib():
    global _lib
    if _lib is None:
        from ethereum_consensus_amd import build
        path = build.build_hostsim(verbose=False)
        _lib = ctypes.CDLL(path)
        _lib.hs_pass.restype = ctypes.c_uint64
        _lib.hs_pass.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                 ctypes.c_void_p, ctypes.c_int]
        _lib.hs_tree_job.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int,
                                     ctypes.c_uint64, ctypes.c_void_p]
        _lib.hs_sha256.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        _lib.hs_xmd.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p]
        _lib.hs_hash_to_g2.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        _lib.hs_hash_to_g2_split.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        _lib.hs_fast_aggregate_verify.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint64,
                                                  ctypes.c_char_p, ctypes.c_int]
    return _lib


def unaligned_buffer(data: bytes, misalign: int):
    """ctypes buffer whose payload starts `misalign` bytes past a 16-byte boundary, with guard bytes."""
    raw = ctypes.create_string_buffer(len(data) + 64)
    base = ctypes.addressof(raw)
    start = (-base) % 16 + misalign
    ctypes.memmove(base + start, data, len(data))
    # poison the bytes after the payload: the lane programs must never let them leak in
    for i in range(start + len(data), len(raw)):
        raw[i] = b"\xa5"
    return raw, base + start


def merkleize(kind: int, data: bytes, n0: int, depth: int, mix: bool, mix_len: int, ds, misalign: int = 0) -> bytes:
    """Compose passes with the given per-pass heights `ds` then the finishing job."""
    L = lib()
    raw, addr = unaligned_buffer(data, misalign)
    n, level, first = n0, 0, True
    cur = None
    for D in ds:
        if n == 0:
            break
        D = min(D, depth - level)
        n_out = (n + (1 << D) - 1) >> D
        out = ctypes.create_string_buffer(32 * max(n_out, 1))
        if first:
            L.hs_pass(kind, D, addr, len(data), n, out, 0)
        else:
            L.hs_pass(1, D, cur, 32 * n, n, out, level)
        cur, n, level, first = out, n_out, level + D, False
    res = ctypes.create_string_buffer(32)
    if first and n > 0:
        raise ValueError("need at least one pass for a leaf functor")
    L.hs_tree_job(cur if cur is not None else res, n, level, depth, int(mix), mix_len, res)
    return res.raw


def merkleize_scheduled(kind: int, data: bytes, n0: int, depth: int, mix: bool, mix_len: int, misalign: int = 0):
    """hs_merkleize: the product's own pass schedule (schedule_merkleize) on the lane simulator."""
    L = lib()
    L.hs_merkleize.restype = ctypes.c_uint64
    L.hs_merkleize.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                               ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p]
    raw, addr = unaligned_buffer(data, misalign)
    out = ctypes.create_string_buffer(32)
    h = L.hs_merkleize(kind, addr, len(data), n0, depth, int(mix), mix_len, out)
    return out.raw, h


def state_root_deneb(ssz: bytes, preset: int):
    L = lib()
    L.hs_state_root_deneb.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.POINTER(ctypes.c_uint64)]
    out = ctypes.create_string_buffer(32)
    h = ctypes.c_uint64(0)
    rc = L.hs_state_root_deneb(ssz, len(ssz), preset, out, ctypes.byref(h))
    return rc, out.raw, h.value


def state_root_fork(fork: int, ssz: bytes, preset: int):
    L = lib()
    L.hs_state_root_fork.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.POINTER(ctypes.c_uint64)]
    out = ctypes.create_string_buffer(32)
    h = ctypes.c_uint64(0)
    rc = L.hs_state_root_fork(fork, ssz, len(ssz), preset, out, ctypes.byref(h))
    return rc, out.raw, h.value
