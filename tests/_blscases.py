"""Shared BLS test material: seeded key/signature sets, malformed encodings and the edge cases of the
reference wrappers (/root/reference/ethereum-consensus/src/crypto/bls.rs:64-160), each paired with the
status the oracle (oracle/bls12_381.py) assigns.  Used by the CPU (hostsim) and GPU parity tests."""
import random

from oracle import bls12_381 as B

# crypto/bls.rs:530-544 `test_can_sign`
CAN_SIGN_SK = int("40094c5c6c378857eac09b8ec64c87182f58700c056a8b371ad0eb0a5b983d50", 16)
CAN_SIGN_MSG = b"blst is such a blast"
CAN_SIGN_SIG = bytes.fromhex(
    "a01e49276730e4752eef31b0570c8707de501398dac70dd144438cd1bd05fb9b9bb3e1a9ceef0a68cc08904362cafa3f"
    "1005e5b699a41847fff6f5552260468846de5bdbf94a9aedeb29bc6cdb2c1d34922d9e9af4c0593a69ae978a90b5aba6")
# bin/ec/validator/keystores.rs:240-249 (EIP-2335)
EIP2335_SK = int("000000000019d6689c085ae165831e934ff763ae46a2a6c172b3f1b60a8ce26f", 16)
EIP2335_PK = bytes.fromhex(
    "9612d7a727c9d0a22e185a1c768478dfe919cada9266988cb32359c11f2b7b27f4ae4040902382ae2910c15e2b420d07")


def rand_g1_curve_point(r):
    """on E1, (almost surely) outside G1"""
    while True:
        x = r.randrange(B.P)
        y = B.fp_sqrt((x ** 3 + 4) % B.P)
        if y is not None:
            return (x, y)


def rand_g2_curve_point(r):
    while True:
        x = (r.randrange(B.P), r.randrange(B.P))
        y = B.f2_sqrt(B.f2_add(B.f2_mul(B.f2_sqr(x), x), B.B2))
        if y is not None:
            return (x, y)


def malformed_g1(r):
    """48-byte strings exercising every decode branch"""
    out = [B.INFINITY_PUBLIC_KEY, bytes([0x80]) + bytes(47), bytes([0xA0]) + bytes(47), bytes(48),
           bytes([0xC0]) + bytes(46) + b"\x01", bytes([0xE0]) + bytes(47), bytes([0x9F]) + b"\xff" * 47]
    pb = bytearray(B.P.to_bytes(48, "big"))
    pb[0] |= 0x80
    out.append(bytes(pb))
    pm = bytearray((B.P - 1).to_bytes(48, "big"))
    pm[0] |= 0x80
    out.append(bytes(pm))
    for i in range(6):
        b = bytearray(r.randbytes(48))
        b[0] = (b[0] & 0x0F) | 0x80 | (0x20 if i & 1 else 0)
        out.append(bytes(b))
    return out


def malformed_g2(r):
    out = [B.INFINITY_SIGNATURE, bytes(96), bytes([0x80]) + bytes(95), bytes([0xC0]) + bytes(94) + b"\x01",
           bytes([0x9F]) + b"\xff" * 95]
    x0b = bytearray(bytes(48) + B.P.to_bytes(48, "big"))
    x0b[0] |= 0x80
    out.append(bytes(x0b))
    for i in range(6):
        b = bytearray(r.randbytes(96))
        b[0] = (b[0] & 0x0F) | 0x80 | (0x20 if i & 1 else 0)
        b[48] &= 0x0F
        out.append(bytes(b))
    return out


def fav_cases(seed=9):
    """[(pks, msg, sig, eth_variant)] covering the status algebra of (eth_)fast_aggregate_verify."""
    r = random.Random(seed)
    sks = [r.randrange(1, B.R) for _ in range(4)]
    pks = [B.sk_to_pk(s) for s in sks]
    msg = r.randbytes(32)
    H = B.hash_to_g2(msg)
    sigs = [B.g2_compress(B.g2_mul(H, s)) for s in sks]
    agg = B.g2_compress(B.g2_mul(H, sum(sks) % B.R))
    neg0 = B.g1_compress(B.g1_neg(B.g1_decompress(pks[0])[1]))
    off_g1 = B.g1_compress(rand_g1_curve_point(r))
    off_g2 = B.g2_compress(rand_g2_curve_point(r))
    cases = [
        (pks[:1], msg, sigs[0], 0), (pks, msg, agg, 0), (pks, msg, sigs[0], 0), (pks[:1], msg + b"x", sigs[0], 0),
        (pks[:2], msg, agg, 0), ([], msg, sigs[0], 0), ([], msg, B.INFINITY_SIGNATURE, 0),
        ([], msg, B.INFINITY_SIGNATURE, 1), ([], msg, sigs[0], 1), (pks[:1], msg, B.INFINITY_SIGNATURE, 0),
        (pks[:1], msg, B.INFINITY_SIGNATURE, 1), ([B.INFINITY_PUBLIC_KEY], msg, sigs[0], 0),
        ([pks[0], B.INFINITY_PUBLIC_KEY], msg, sigs[0], 0), ([pks[0], neg0], msg, sigs[0], 0),
        ([pks[0], bytes(48)], msg, bytes(96), 0), ([pks[0]], msg, bytes(96), 0),
        (pks[:1], msg, off_g2, 0), ([off_g1], msg, sigs[0], 0), ([pks[0], off_g1], msg, off_g2, 0),
        ([pks[1], pks[1]], msg, B.g2_compress(B.g2_mul(H, 2 * sks[1] % B.R)), 0),  # repeated key: doubling
        ([B.sk_to_pk(CAN_SIGN_SK)], CAN_SIGN_MSG, CAN_SIGN_SIG, 0),
        ([B.sk_to_pk(CAN_SIGN_SK)], CAN_SIGN_MSG, CAN_SIGN_SIG, 1),
    ]
    return cases


def oracle_fav(pks, msg, sig, eth):
    return B.eth_fast_aggregate_verify(pks, msg, sig) if eth else B.fast_aggregate_verify(pks, msg, sig)
