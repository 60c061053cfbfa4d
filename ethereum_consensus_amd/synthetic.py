"""Deterministic synthetic inputs for tests and bench.py (pure data construction, no hashing).

`beacon_state_deneb(n, preset)` builds the SSZ encoding of a deneb BeaconState
(/root/reference/ethereum-consensus/src/deneb/beacon_state.rs:13-64) with `n` validators in the
shape SURVEY.md 8(d) config 3 describes: mostly-active validators with 32 ETH effective
balance, FAR_FUTURE exit epochs for 15/16 of the entries, random roots/mixes, full-size
vectors, empty-ish history lists.  numpy only, so 2^20 validators take well under a second.
"""
from __future__ import annotations

import numpy as np

FAR_FUTURE_EPOCH = 2**64 - 1

PRESETS = {
    "mainnet": dict(id=0, SLOTS_PER_HISTORICAL_ROOT=8192, EPOCHS_PER_HISTORICAL_VECTOR=65536,
                    EPOCHS_PER_SLASHINGS_VECTOR=8192, SYNC_COMMITTEE_SIZE=512, ETH1_DATA_VOTES_BOUND=2048),
    "minimal": dict(id=1, SLOTS_PER_HISTORICAL_ROOT=64, EPOCHS_PER_HISTORICAL_VECTOR=64,
                    EPOCHS_PER_SLASHINGS_VECTOR=64, SYNC_COMMITTEE_SIZE=32, ETH1_DATA_VOTES_BOUND=32),
}

VALIDATOR_DTYPE = np.dtype([
    ("public_key", "V48"), ("withdrawal_credentials", "V32"), ("effective_balance", "<u8"), ("slashed", "u1"),
    ("activation_eligibility_epoch", "<u8"), ("activation_epoch", "<u8"), ("exit_epoch", "<u8"),
    ("withdrawable_epoch", "<u8")])
assert VALIDATOR_DTYPE.itemsize == 121


def validators(n: int, seed: int = 1) -> np.ndarray:
    rng = np.random.default_rng(seed)
    v = np.zeros(n, dtype=VALIDATOR_DTYPE)
    raw = v.view(np.uint8).reshape(n, 121)
    raw[:, 0:48] = rng.integers(0, 256, size=(n, 48), dtype=np.uint8)
    raw[:, 48] = 1  # 0x01 withdrawal prefix, 11 zero bytes, 20 address bytes
    raw[:, 60:80] = rng.integers(0, 256, size=(n, 20), dtype=np.uint8)
    i = np.arange(n, dtype=np.uint64)
    v["effective_balance"] = np.where(i % 16 == 3, 31 * 10**9, 32 * 10**9).astype(np.uint64)
    v["slashed"] = (i % 1024 == 0).astype(np.uint8)
    ep = rng.integers(0, 1 << 18, size=(n, 4), dtype=np.uint64)
    v["activation_eligibility_epoch"] = ep[:, 0]
    v["activation_epoch"] = ep[:, 1]
    exited = (i % 16 == 7)
    v["exit_epoch"] = np.where(exited, ep[:, 2], np.uint64(FAR_FUTURE_EPOCH))
    v["withdrawable_epoch"] = np.where(exited, ep[:, 3], np.uint64(FAR_FUTURE_EPOCH))
    return v


def state_fields(n: int, preset: str = "mainnet", seed: int = 1, n_votes: int = 5, n_hist_roots: int = 3,
                 n_hist_summaries: int = 4, extra_data: bytes = b"ecgpu") -> dict:
    P = PRESETS[preset]
    rng = np.random.default_rng(seed + 1000)
    rb = lambda *shape: rng.integers(0, 256, size=shape, dtype=np.uint8)
    i = np.arange(n, dtype=np.uint64)
    f = dict(
        genesis_time=1606824023, genesis_validators_root=rb(32).tobytes(), slot=8_626_176,
        fork=(b"\x03\x00\x00\x00", b"\x04\x00\x00\x00", 269568),
        latest_block_header=(8_626_175, 123456, rb(32).tobytes(), rb(32).tobytes(), rb(32).tobytes()),
        block_roots=rb(P["SLOTS_PER_HISTORICAL_ROOT"], 32), state_roots=rb(P["SLOTS_PER_HISTORICAL_ROOT"], 32),
        historical_roots=rb(n_hist_roots, 32),
        eth1_data=(rb(32).tobytes(), 1_000_000, rb(32).tobytes()),
        eth1_data_votes=[(rb(32).tobytes(), 1_000_000 + k, rb(32).tobytes()) for k in range(n_votes)],
        eth1_deposit_index=1_000_000,
        validators=validators(n, seed),
        balances=(np.uint64(32 * 10**9) + rng.integers(0, 10**9, size=n, dtype=np.uint64)).astype("<u8"),
        randao_mixes=rb(P["EPOCHS_PER_HISTORICAL_VECTOR"], 32),
        slashings=rng.integers(0, 1 << 40, size=P["EPOCHS_PER_SLASHINGS_VECTOR"], dtype=np.uint64).astype("<u8"),
        previous_epoch_participation=(rb(n) & 7), current_epoch_participation=(rb(n) & 7),
        justification_bits=0b1011,
        previous_justified_checkpoint=(269566, rb(32).tobytes()),
        current_justified_checkpoint=(269567, rb(32).tobytes()),
        finalized_checkpoint=(269566, rb(32).tobytes()),
        inactivity_scores=np.where(i % 4096 == 0, 7, 0).astype("<u8"),
        current_sync_committee=(rb(P["SYNC_COMMITTEE_SIZE"], 48), rb(48).tobytes()),
        next_sync_committee=(rb(P["SYNC_COMMITTEE_SIZE"], 48), rb(48).tobytes()),
        payload_header=dict(
            parent_hash=rb(32).tobytes(), fee_recipient=rb(20).tobytes(), state_root=rb(32).tobytes(),
            receipts_root=rb(32).tobytes(), logs_bloom=rb(256).tobytes(), prev_randao=rb(32).tobytes(),
            block_number=19_000_000, gas_limit=30_000_000, gas_used=12_345_678, timestamp=1_710_000_000,
            extra_data=bytes(extra_data), base_fee_per_gas=23_000_000_000, block_hash=rb(32).tobytes(),
            transactions_root=rb(32).tobytes(), withdrawals_root=rb(32).tobytes(), blob_gas_used=393216,
            excess_blob_gas=786432),
        next_withdrawal_index=40_000_000, next_withdrawal_validator_index=n // 2,
        historical_summaries=rb(n_hist_summaries, 64),
    )
    return f


def _u64(x): return int(x).to_bytes(8, "little")


def serialize_payload_header(h: dict) -> bytes:
    fixed = (h["parent_hash"] + h["fee_recipient"] + h["state_root"] + h["receipts_root"] + h["logs_bloom"]
             + h["prev_randao"] + _u64(h["block_number"]) + _u64(h["gas_limit"]) + _u64(h["gas_used"])
             + _u64(h["timestamp"]) + (584).to_bytes(4, "little") + int(h["base_fee_per_gas"]).to_bytes(32, "little")
             + h["block_hash"] + h["transactions_root"] + h["withdrawals_root"] + _u64(h["blob_gas_used"])
             + _u64(h["excess_blob_gas"]))
    assert len(fixed) == 584
    return fixed + h["extra_data"]


def serialize_state(f: dict) -> bytes:
    """SSZ encoding of the deneb BeaconState described by `f` (field order of the reference)."""
    var = [
        f["historical_roots"].tobytes(),
        b"".join(a + _u64(b) + c for a, b, c in f["eth1_data_votes"]),
        f["validators"].tobytes(), f["balances"].tobytes(), f["previous_epoch_participation"].tobytes(),
        f["current_epoch_participation"].tobytes(), f["inactivity_scores"].tobytes(),
        serialize_payload_header(f["payload_header"]), f["historical_summaries"].tobytes(),
    ]
    OFF = object()
    hdr = f["latest_block_header"]
    parts = [
        _u64(f["genesis_time"]), f["genesis_validators_root"], _u64(f["slot"]),
        f["fork"][0] + f["fork"][1] + _u64(f["fork"][2]),
        _u64(hdr[0]) + _u64(hdr[1]) + hdr[2] + hdr[3] + hdr[4],
        f["block_roots"].tobytes(), f["state_roots"].tobytes(), OFF,
        f["eth1_data"][0] + _u64(f["eth1_data"][1]) + f["eth1_data"][2], OFF, _u64(f["eth1_deposit_index"]),
        OFF, OFF, f["randao_mixes"].tobytes(), f["slashings"].tobytes(), OFF, OFF,
        bytes([f["justification_bits"]]),
        _u64(f["previous_justified_checkpoint"][0]) + f["previous_justified_checkpoint"][1],
        _u64(f["current_justified_checkpoint"][0]) + f["current_justified_checkpoint"][1],
        _u64(f["finalized_checkpoint"][0]) + f["finalized_checkpoint"][1], OFF,
        f["current_sync_committee"][0].tobytes() + f["current_sync_committee"][1],
        f["next_sync_committee"][0].tobytes() + f["next_sync_committee"][1], OFF,
        _u64(f["next_withdrawal_index"]), _u64(f["next_withdrawal_validator_index"]), OFF,
    ]
    fixed_len = sum(4 if p is OFF else len(p) for p in parts)
    out, off, vi = [], fixed_len, 0
    for p in parts:
        if p is OFF:
            out.append(off.to_bytes(4, "little"))
            off += len(var[vi])
            vi += 1
        else:
            out.append(p)
    return b"".join(out) + b"".join(var)


def beacon_state_deneb(n: int, preset: str = "mainnet", seed: int = 1, **kw) -> bytes:
    return serialize_state(state_fields(n, preset, seed, **kw))


def expected_hash64_count(n: int, preset: str = "mainnet") -> int:
    """hash64 evaluations of one whole-state root: bookkeeping for bench.py (mirrors the SSZ
    rules; the library reports its own count through ecgpu_last_hash64_count)."""
    raise NotImplementedError("use ecgpu_last_hash64_count()")


# ---------------------------------------------------------------------------------------------------------------------
# BLS workload of SURVEY.md 8(d) config 2: B independent K = 1 tuples with fault injection at i = 0 (mod 64), cycling
# through eight fault classes.  Pure data construction on compressed encodings (ZCash format, crypto/bls.rs:227-349);
# keys and signatures themselves come from the device (ecgpu_sk_to_pk_batch / ecgpu_sign_batch).
# ---------------------------------------------------------------------------------------------------------------------
BLS_P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
BLS_R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
INFINITY_PUBLIC_KEY = bytes([0xC0]) + bytes(47)
INFINITY_SIGNATURE = bytes([0xC0]) + bytes(95)
# statuses by construction (include/ecgpu.h): wrong message, swapped key, signature outside G2 (found by verify's group
# check), key outside G1, compression flag cleared, x >= p, key = infinity, signature = infinity
BLS_FAULT_STATUS = (5, 5, 0x43, 3, 1, 1, 6, 5)
BLS_FAULT_NAMES = ("wrong message", "swapped public key", "signature outside the G2 subgroup", "public key outside the G1 subgroup",
                   "compression flag cleared", "x >= p", "public key = infinity", "signature = infinity")


def bls_tag(tag: bytes, i: int) -> bytes:
    """S(tag, i) = SHA-256("ecgpu/v1/" || tag || "/" || u32le(i)) of SURVEY.md 8(d)."""
    import hashlib
    return hashlib.sha256(b"ecgpu/v1/" + tag + b"/" + int(i).to_bytes(4, "little")).digest()


def bls_secret_keys(n: int, base: int = 0) -> bytes:
    """sk_i = 1 + S("sk", i) mod (r - 1), 32 big-endian bytes each"""
    return b"".join((1 + int.from_bytes(bls_tag(b"sk", base + i), "big") % (BLS_R - 1)).to_bytes(32, "big") for i in range(n))


def bls_messages(n: int, base: int = 0, tag: bytes = b"msg") -> bytes:
    return b"".join(bls_tag(tag, base + i) for i in range(n))


def _fp_sqrt(a: int):
    s = pow(a, (BLS_P + 1) // 4, BLS_P)
    return s if s * s % BLS_P == a % BLS_P else None


def _f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % BLS_P, (a[0] * b[1] + a[1] * b[0]) % BLS_P)


def _f2_sqrt(a):
    if a[1] == 0:
        s = _fp_sqrt(a[0])
        if s is not None:
            return (s, 0)
        s = _fp_sqrt(-a[0] % BLS_P)
        return None if s is None else (0, s)
    n = _fp_sqrt((a[0] * a[0] + a[1] * a[1]) % BLS_P)
    if n is None:
        return None
    inv2 = pow(2, BLS_P - 2, BLS_P)
    for cand in ((a[0] + n) * inv2 % BLS_P, (a[0] - n) * inv2 % BLS_P):
        x0 = _fp_sqrt(cand)
        if x0:
            x1 = a[1] * pow(2 * x0, BLS_P - 2, BLS_P) % BLS_P
            if _f2_mul((x0, x1), (x0, x1)) == (a[0] % BLS_P, a[1] % BLS_P):
                return (x0, x1)
    return None


def off_subgroup_public_key(i: int) -> bytes:
    """a compressed point ON E1: y^2 = x^3 + 4 that is (with probability 1 - 2^-125) outside G1"""
    x = int.from_bytes(bls_tag(b"offg1", i) + bls_tag(b"offg1b", i)[:16], "big") % BLS_P
    while True:
        y = _fp_sqrt((x * x * x + 4) % BLS_P)
        if y is not None:
            break
        x = (x + 1) % BLS_P
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if y > (BLS_P - 1) // 2 else 0)
    return bytes(b)


def off_subgroup_signature(i: int) -> bytes:
    """a compressed point ON E2: y^2 = x^3 + 4 (1 + i) that is (almost surely) outside G2"""
    x0 = int.from_bytes(bls_tag(b"offg2", i) + bls_tag(b"offg2b", i)[:16], "big") % BLS_P
    x1 = int.from_bytes(bls_tag(b"offg2c", i) + bls_tag(b"offg2d", i)[:16], "big") % BLS_P
    while True:
        x = (x0, x1)
        x3 = _f2_mul(_f2_mul(x, x), x)
        y = _f2_sqrt(((x3[0] + 4) % BLS_P, (x3[1] + 4) % BLS_P))
        if y is not None:
            break
        x0 = (x0 + 1) % BLS_P
    largest = y[1] > (BLS_P - 1) // 2 if y[1] else y[0] > (BLS_P - 1) // 2
    b = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if largest else 0)
    return bytes(b)


def bls_inject_faults(pks: bytearray, msgs: bytearray, sigs: bytearray, n: int, period: int = 64, n_distinct: int = 4):
    """Corrupt tuple i for every i = 0 (mod period), fault class (i / period) mod 8, in place.
    Returns the n status bytes the reference semantics assign (include/ecgpu.h numbering) and the fault class per tuple
    (255 = untouched).  A `swapped key` takes the key of tuple i + 1 (wraps to i - 1 at the end)."""
    want = bytearray(n)
    kind_of = bytearray([255]) * n
    off1 = [off_subgroup_public_key(j) for j in range(n_distinct)]
    off2 = [off_subgroup_signature(j) for j in range(n_distinct)]
    for i in range(0, n, period):
        kind = (i // period) % 8
        kind_of[i] = kind
        want[i] = BLS_FAULT_STATUS[kind]
        if kind == 0:
            msgs[32 * i] ^= 1
        elif kind == 1:
            j = i + 1 if i + 1 < n else i - 1
            pks[48 * i:48 * i + 48] = pks[48 * j:48 * j + 48]
        elif kind == 2:
            sigs[96 * i:96 * i + 96] = off2[(i // period // 8) % n_distinct]
        elif kind == 3:
            pks[48 * i:48 * i + 48] = off1[(i // period // 8) % n_distinct]
        elif kind == 4:
            pks[48 * i] &= 0x7F
        elif kind == 5:
            sigs[96 * i:96 * i + 48] = bytes([0x9F]) + b"\xff" * 47
        elif kind == 6:
            pks[48 * i:48 * i + 48] = INFINITY_PUBLIC_KEY
        else:
            sigs[96 * i:96 * i + 96] = INFINITY_SIGNATURE
    return want, kind_of
