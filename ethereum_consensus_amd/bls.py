"""Host-side mirror of `ethereum_consensus::crypto` (BLS half), backed by the HIP kernels.

Same names, argument meaning and error behaviour as the reference wrappers in
/root/reference/ethereum-consensus/src/crypto/bls.rs:
    verify_signature :64-77, aggregate :79-93, aggregate_verify :95-112,
    fast_aggregate_verify :114-132, eth_aggregate_public_keys :135-148,
    eth_fast_aggregate_verify :150-160, Error :27-42, BLSTError :44-62.
Public keys are 48 bytes, signatures 96 bytes (`ByteVector<48|96>`, bls.rs:239,290): any other length
is rejected here exactly as the reference's constructors reject it (bls.rs:372-406,463-487).
Everything is computed on the GPU through libecgpu.so; a missing library or device raises.
The `*_batch` functions are the extension the GPU backend exists for: one call per block / epoch.
"""
from __future__ import annotations

import ctypes
from typing import Sequence

from . import _lib

BLS_DST = b"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_"  # bls.rs:22
PUBLIC_KEY_BYTES_LEN = 48
SIGNATURE_BYTES_LEN = 96
INFINITY_COMPRESSED_SIGNATURE = bytes([0xC0]) + bytes(95)  # bls.rs:338-343

_BLST_STRINGS = {  # bls.rs:48-62
    1: "bad encoding",
    2: "point not on curve",
    3: "point not in group",
    4: "aggregation type mismatch",
    5: "verification failed",
    6: "public key is infinity",
    7: "bad scalar",
}
# BLST_ERRORs of the conversions TryFrom<&PublicKey> / TryFrom<&Signature> -> Error::BLST (bls.rs:69-70,100-105,119-125).
# The same two group / infinity conditions met INSIDE blst's verify call come back as 0x43 / 0x46 (ECGPU_IN_VERIFY set,
# include/ecgpu.h) and, like every other non-zero result of that call, are Error::InvalidSignature (bls.rs:72-76).
_CONVERSION_CODES = (1, 2, 3, 6)
IN_VERIFY = 0x40
EMPTY_AGGREGATE = -100


class Error(Exception):
    """crypto::Error (bls.rs:27-42)."""


class InvalidSignature(Error):
    def __init__(self):
        super().__init__("invalid signature")


class EmptyAggregate(Error):
    def __init__(self):
        super().__init__("attempt to aggregate empty set")


class BLSTError(Error):
    def __init__(self, code: int):
        self.code = code
        super().__init__(_BLST_STRINGS.get(code, f"unknown blst error {code}"))


class InvalidLength(Error):
    """ByteVector<N> construction failure (ssz/byte_vector.rs:24-30)."""


def _pk(b: bytes) -> bytes:
    b = bytes(b)
    if len(b) != PUBLIC_KEY_BYTES_LEN:
        raise InvalidLength(f"public key must be {PUBLIC_KEY_BYTES_LEN} bytes, got {len(b)}")
    return b


def _sig(b: bytes) -> bytes:
    b = bytes(b)
    if len(b) != SIGNATURE_BYTES_LEN:
        raise InvalidLength(f"signature must be {SIGNATURE_BYTES_LEN} bytes, got {len(b)}")
    return b


def _buf(b: bytes):
    return ctypes.create_string_buffer(bytes(b), len(b)) if len(b) else ctypes.create_string_buffer(1)


def _raise_for(code: int) -> None:
    """Map a status of the verify functions to the reference's Result: conversion errors -> Error::BLST, whatever blst's
    verify call returned -> InvalidSignature."""
    if code == 0:
        return
    if code in _CONVERSION_CODES:
        raise BLSTError(code)
    raise InvalidSignature()


def _raise_for_aggregate(code: int) -> None:
    """aggregate / eth_aggregate_public_keys: EmptyAggregate, else every BLST_ERROR is Error::BLST (bls.rs:92,147)."""
    if code == 0:
        return
    if code == EMPTY_AGGREGATE:
        raise EmptyAggregate()
    raise BLSTError(code & 7)


# ---- status-returning layer (what the C ABI returns; used by the parity tests) -------------------
def verify_signature_status(public_key: bytes, msg: bytes, signature: bytes) -> int:
    L = _lib.load()
    return _lib.check(L.ecgpu_verify(_buf(_pk(public_key)), _buf(msg), len(msg), _buf(_sig(signature))), "ecgpu_verify")


def fast_aggregate_verify_status(public_keys: Sequence[bytes], msg: bytes, signature: bytes, eth: bool = False) -> int:
    L = _lib.load()
    pks = b"".join(_pk(p) for p in public_keys)
    return _lib.check(L.ecgpu_fast_aggregate_verify(_buf(pks), len(public_keys), _buf(msg), len(msg), _buf(_sig(signature)),
                                                    1 if eth else 0), "ecgpu_fast_aggregate_verify")


def aggregate_verify_status(public_keys: Sequence[bytes], msgs: Sequence[bytes], signature: bytes) -> int:
    L = _lib.load()
    pks = b"".join(_pk(p) for p in public_keys)
    off = [0]
    for m in msgs:
        off.append(off[-1] + len(m))
    off_arr = (ctypes.c_uint64 * len(off))(*off)
    return _lib.check(L.ecgpu_aggregate_verify(_buf(pks), len(public_keys), _buf(b"".join(msgs)), off_arr, len(msgs),
                                               _buf(_sig(signature))), "ecgpu_aggregate_verify")


def aggregate_status(signatures: Sequence[bytes]):
    L = _lib.load()
    out = ctypes.create_string_buffer(96)
    rc = L.ecgpu_aggregate_sigs(_buf(b"".join(_sig(s) for s in signatures)), len(signatures), out)
    _lib.check(rc, "ecgpu_aggregate_sigs")
    return rc, (out.raw if rc == 0 else None)


def eth_aggregate_public_keys_status(public_keys: Sequence[bytes]):
    L = _lib.load()
    out = ctypes.create_string_buffer(48)
    rc = L.ecgpu_aggregate_pks(_buf(b"".join(_pk(p) for p in public_keys)), len(public_keys), out)
    _lib.check(rc, "ecgpu_aggregate_pks")
    return rc, (out.raw if rc == 0 else None)


# ---- the reference's API ----------------------------------------------------------------------------
def verify_signature(public_key: bytes, msg: bytes, signature: bytes) -> None:
    """bls.rs:64-77: Ok(()) or Err."""
    _raise_for(verify_signature_status(public_key, msg, signature))


def aggregate(signatures: Sequence[bytes]) -> bytes:
    """bls.rs:79-93."""
    rc, out = aggregate_status(signatures)
    _raise_for_aggregate(rc)
    return out


def aggregate_verify(public_keys: Sequence[bytes], msgs: Sequence[bytes], signature: bytes) -> None:
    """bls.rs:95-112."""
    _raise_for(aggregate_verify_status(public_keys, msgs, signature))


def fast_aggregate_verify(public_keys: Sequence[bytes], msg: bytes, signature: bytes) -> None:
    """bls.rs:114-132."""
    _raise_for(fast_aggregate_verify_status(public_keys, msg, signature))


def eth_aggregate_public_keys(public_keys: Sequence[bytes]) -> bytes:
    """bls.rs:135-148."""
    rc, out = eth_aggregate_public_keys_status(public_keys)
    _raise_for_aggregate(rc)
    return out


def eth_fast_aggregate_verify(public_keys: Sequence[bytes], msg: bytes, signature: bytes) -> None:
    """bls.rs:150-160."""
    _raise_for(fast_aggregate_verify_status(public_keys, msg, signature, eth=True))


# ---- batch extension + SecretKey side ------------------------------------------------------------------
def fast_aggregate_verify_batch(public_keys: bytes, pk_offsets, msgs32: bytes, signatures: bytes, eth: bool = False) -> bytes:
    """n independent fast_aggregate_verify over 32-byte messages.  `public_keys` = concatenated 48-byte
    keys; tuple i uses keys pk_offsets[i]..pk_offsets[i+1] (None: one key per tuple).  Returns n status
    bytes (BLST_ERROR numbering), each what the scalar call would have produced."""
    L = _lib.load()
    n = len(signatures) // 96
    if len(signatures) != 96 * n or len(msgs32) != 32 * n or len(public_keys) % 48:
        raise InvalidLength("batch buffers must hold 96-byte signatures, 32-byte messages, 48-byte keys")
    off = None
    if pk_offsets is not None:
        if len(pk_offsets) != n + 1 or pk_offsets[-1] * 48 != len(public_keys):
            raise InvalidLength("pk_offsets must have n+1 entries ending at the key count")
        off = (ctypes.c_uint32 * (n + 1))(*pk_offsets)
    elif len(public_keys) != 48 * n:
        raise InvalidLength("one key per tuple expected")
    out = ctypes.create_string_buffer(max(n, 1))
    _lib.check(L.ecgpu_fast_aggregate_verify_batch(_buf(public_keys), off, _buf(msgs32), _buf(signatures), n, 1 if eth else 0,
                                                   out), "ecgpu_fast_aggregate_verify_batch")
    return out.raw[:n]


class ValidatorKeyRegistry:
    """Device-resident result of `PublicKey -> blst key` (bls.rs:279-285) per validator index: the affine point, or
    the BLSTError the conversion raises.  `fast_aggregate_verify_batch(indices...)` then returns exactly what the
    reference returns for the same keys without decompressing them again.  `set` is the hook
    for `add_validator_to_registry` (phase0/block_processing.rs:317-349)."""

    def __init__(self, capacity: int):
        self._L = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(self._L.ecgpu_registry_create(capacity, ctypes.byref(h)), "ecgpu_registry_create")
        self._h = h
        self.capacity = capacity

    def close(self):
        if self._h:
            self._L.ecgpu_registry_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def handle(self):
        return self._h

    def set(self, first_index: int, public_keys: bytes) -> None:
        if len(public_keys) % 48:
            raise InvalidLength("48-byte keys expected")
        _lib.check(self._L.ecgpu_registry_set(self._h, first_index, _buf(public_keys), len(public_keys) // 48), "ecgpu_registry_set")

    def fast_aggregate_verify_batch(self, indices: Sequence[int], idx_offsets: Sequence[int], msgs32: bytes, signatures: bytes,
                                    eth: bool = False) -> bytes:
        n = len(signatures) // 96
        if len(signatures) != 96 * n or len(msgs32) != 32 * n or len(idx_offsets) != n + 1 or idx_offsets[-1] != len(indices):
            raise InvalidLength("n signatures, n messages, n+1 offsets ending at the index count")
        idx = (ctypes.c_uint32 * max(len(indices), 1))(*indices)
        off = (ctypes.c_uint32 * (n + 1))(*idx_offsets)
        out = ctypes.create_string_buffer(max(n, 1))
        _lib.check(self._L.ecgpu_fast_aggregate_verify_indexed_batch(self._h, idx, off, _buf(msgs32), _buf(signatures), n,
                                                                     1 if eth else 0, out), "ecgpu_fast_aggregate_verify_indexed_batch")
        return out.raw[:n]


def sk_to_pk_batch(secret_keys32: bytes) -> bytes:
    """SecretKey::public_key (bls.rs:193-197) for n 32-byte big-endian secret keys."""
    L = _lib.load()
    n = len(secret_keys32) // 32
    out = ctypes.create_string_buffer(max(48 * n, 1))
    _lib.check(L.ecgpu_sk_to_pk_batch(_buf(secret_keys32), n, out), "ecgpu_sk_to_pk_batch")
    return out.raw[:48 * n]


def sign_batch(secret_keys32: bytes, msgs: Sequence[bytes]) -> bytes:
    """SecretKey::sign (bls.rs:213-219): sig_i = [sk_i] hash_to_G2(msg_i)."""
    L = _lib.load()
    n = len(secret_keys32) // 32
    assert n == len(msgs)
    off = [0]
    for m in msgs:
        off.append(off[-1] + len(m))
    off_arr = (ctypes.c_uint64 * len(off))(*off)
    out = ctypes.create_string_buffer(max(96 * n, 1))
    _lib.check(L.ecgpu_sign_batch(_buf(secret_keys32), _buf(b"".join(msgs)), off_arr, n, out), "ecgpu_sign_batch")
    return out.raw[:96 * n]


class SignatureBatch:
    """Whole-block batching: queue every verification of a block -- `verify_signature`,
    `fast_aggregate_verify`, `eth_fast_aggregate_verify` -- and verify them in ONE pass of the GPU pipeline.  `flush()`
    returns, per queued call in order, the status the scalar call would have returned; `results()` maps them to the
    reference's Result (None = Ok, or the Error instance the call would have raised).  Call sites it collects:
    phase0/state_transition.rs:56, phase0/block_processing.rs:649,752-761, altair/block_processing.rs:226-234."""

    def __init__(self, registry: "ValidatorKeyRegistry | None" = None):
        self._L = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(self._L.ecgpu_batch_create(registry.handle if registry is not None else None, ctypes.byref(h)), "ecgpu_batch_create")
        self._h = h
        self._reg = registry  # keeps the registry alive

    def close(self):
        if self._h:
            self._L.ecgpu_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __len__(self):
        return int(self._L.ecgpu_batch_len(self._h))

    def _pushed(self, rc: int) -> int:
        if rc < 0:
            raise _lib.EcgpuError(int(rc), "ecgpu_batch_push")
        return int(rc)

    def verify_signature(self, public_key: bytes, msg: bytes, signature: bytes) -> int:
        return self._pushed(self._L.ecgpu_batch_push(self._h, _buf(_pk(public_key)), 1, _buf(msg), len(msg), _buf(_sig(signature)), 0))

    def fast_aggregate_verify(self, public_keys: Sequence[bytes], msg: bytes, signature: bytes, eth: bool = False) -> int:
        pks = b"".join(_pk(p) for p in public_keys)
        return self._pushed(self._L.ecgpu_batch_push(self._h, _buf(pks), len(public_keys), _buf(msg), len(msg), _buf(_sig(signature)),
                                                     1 if eth else 0))

    def eth_fast_aggregate_verify(self, public_keys: Sequence[bytes], msg: bytes, signature: bytes) -> int:
        return self.fast_aggregate_verify(public_keys, msg, signature, eth=True)

    def fast_aggregate_verify_indexed(self, indices: Sequence[int], msg: bytes, signature: bytes, eth: bool = False) -> int:
        idx = (ctypes.c_uint32 * max(len(indices), 1))(*indices)
        return self._pushed(self._L.ecgpu_batch_push_indexed(self._h, idx, len(indices), _buf(msg), len(msg), _buf(_sig(signature)),
                                                             1 if eth else 0))

    def flush(self) -> bytes:
        n = len(self)
        out = ctypes.create_string_buffer(max(n, 1))
        _lib.check(self._L.ecgpu_batch_flush(self._h, out, n), "ecgpu_batch_flush")
        return out.raw[:n]

    def results(self):
        out = []
        for st in self.flush():
            try:
                _raise_for(st)
                out.append(None)
            except Error as e:
                out.append(e)
        return out


def fast_aggregate_verify_batch_multi(devices: Sequence[int], public_keys: bytes, pk_offsets, msgs32: bytes, signatures: bytes,
                                      eth: bool = False) -> bytes:
    """`fast_aggregate_verify_batch` sharded over several GPUs of this process (one host thread per device inside the
    library).  Same statuses as the single-device call."""
    L = _lib.load()
    n = len(signatures) // 96
    off = (ctypes.c_uint32 * (n + 1))(*pk_offsets) if pk_offsets is not None else None
    devs = (ctypes.c_int * len(devices))(*devices)
    out = ctypes.create_string_buffer(max(n, 1))
    _lib.check(L.ecgpu_fast_aggregate_verify_batch_multi(devs, len(devices), _buf(public_keys), off, _buf(msgs32), _buf(signatures), n,
                                                         1 if eth else 0, out), "ecgpu_fast_aggregate_verify_batch_multi")
    return out.raw[:n]


def g1_multi_scalar_mul(public_keys: Sequence[bytes], scalars: Sequence[int], scalar_bits: int = 255) -> bytes:
    """sum_i [k_i] P_i over G1 (compressed in, compressed out); points validated like `PublicKey -> blst key`"""
    L = _lib.load()
    out = ctypes.create_string_buffer(48)
    sc = b"".join(int(k).to_bytes(32, "big") for k in scalars)
    rc = L.ecgpu_g1_msm(_buf(b"".join(_pk(p) for p in public_keys)), _buf(sc), len(public_keys), scalar_bits, out)
    _lib.check(rc, "ecgpu_g1_msm")
    _raise_for_aggregate(rc)
    return out.raw


def g2_multi_scalar_mul(signatures: Sequence[bytes], scalars: Sequence[int], scalar_bits: int = 255) -> bytes:
    """sum_i [k_i] Q_i over G2 (compressed in, compressed out); points decoded and group-checked like `aggregate` does"""
    L = _lib.load()
    out = ctypes.create_string_buffer(96)
    sc = b"".join(int(k).to_bytes(32, "big") for k in scalars)
    rc = L.ecgpu_g2_msm(_buf(b"".join(_sig(s) for s in signatures)), _buf(sc), len(signatures), scalar_bits, out)
    _lib.check(rc, "ecgpu_g2_msm")
    _raise_for_aggregate(rc)
    return out.raw
