"""Pins oracle/bls12_381.py against every fixed BLS vector the reference holds offline
 plus self-consistency of the derived constants."""
import hashlib

import pytest

from oracle import bls12_381 as B


def test_group_order_matches_reference_modulus():
    # /root/reference/ethereum-consensus/src/bin/ec/bls.rs:6-7
    assert B.R == int("73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001", 16)


def test_generators():
    assert B.g1_on_curve(B.G1) and B.g2_on_curve(B.G2)
    assert B.g1_in_subgroup(B.G1) and B.g2_in_subgroup(B.G2)


def test_eip2335_pubkey_kat():
    # bin/ec/validator/keystores.rs:240-249
    sk = int("000000000019d6689c085ae165831e934ff763ae46a2a6c172b3f1b60a8ce26f", 16)
    assert B.sk_to_pk(sk).hex() == (
        "9612d7a727c9d0a22e185a1c768478dfe919cada9266988cb32359c11f2b7b27f4ae4040902382ae2910c15e2b420d07")


CAN_SIGN_SK = int("40094c5c6c378857eac09b8ec64c87182f58700c056a8b371ad0eb0a5b983d50", 16)
CAN_SIGN_MSG = b"blst is such a blast"
CAN_SIGN_SIG = bytes.fromhex(
    "a01e49276730e4752eef31b0570c8707de501398dac70dd144438cd1bd05fb9b9bb3e1a9ceef0a68cc08904362cafa3f"
    "1005e5b699a41847fff6f5552260468846de5bdbf94a9aedeb29bc6cdb2c1d34922d9e9af4c0593a69ae978a90b5aba6")


def test_can_sign_kat():
    # crypto/bls.rs:530-544 -- pins hash-to-G2 with the ETH DST, G2 scalar mul and compression
    assert B.sign(CAN_SIGN_SK, CAN_SIGN_MSG) == CAN_SIGN_SIG
    # and with the definition-level cofactor clearing (h_eff)
    assert B.g2_compress(B.g2_mul(B.hash_to_g2(CAN_SIGN_MSG, fast=False), CAN_SIGN_SK)) == CAN_SIGN_SIG


def test_can_sign_kat_verifies():
    pk = B.sk_to_pk(CAN_SIGN_SK)
    assert pk.hex() == ("a3843eddcff557c1d9cc39b165688a8211979cef3679ef7c79751023dce64396"
                        "f9ae6b86fa7b1fa15b9041d71dde7614")
    assert B.verify_signature(pk, CAN_SIGN_MSG, CAN_SIGN_SIG) == B.BLST_SUCCESS
    assert B.verify_signature(pk, CAN_SIGN_MSG + b"!", CAN_SIGN_SIG) == B.BLST_VERIFY_FAIL
    assert B.fast_aggregate_verify([pk], CAN_SIGN_MSG, CAN_SIGN_SIG) == B.BLST_SUCCESS
    assert B.aggregate_verify([pk], [CAN_SIGN_MSG], CAN_SIGN_SIG) == B.BLST_SUCCESS


def test_reference_decodable_fixtures():
    # crypto/bls.rs:382-395 `test_signature_from_good_bytes`: decodes (on curve)
    sig = bytes.fromhex(
        "abb0124c7574f281a293f4185cad3cb22681d520917ce46665243eacb051000d8bacf75e1451870ca6b3b9e6c9d41a7b"
        "02ead2685a84188a4fafd3825daf6a989625d719ccd2d83a40101f4a453fca62878c890eca622363f9ddb8f367a91e84")
    st, pt = B.sig_from_bytes(sig)
    assert st == 0 and B.g2_on_curve(pt) and B.g2_compress(pt) == sig
    # crypto/bls.rs:447-456 `good_public_key` (ByteVector only, but it is a real key: validate it)
    pk = bytes.fromhex("a99a76ed7796f7be22d5b7e85deeb7c5677e88e511e0b337618f8c4eb61349b4bf2d153f649f7b53359fe8b94a38e44c")
    st, pt = B.key_validate(pk)
    assert st == 0 and B.g1_compress(pt) == pk
    # sepolia sidecar (deneb/blob_sidecar.rs:74-86): a real G1 commitment and a real G2 signature
    kzg = bytes.fromhex("8da04bbe26b2bbc6b042f4db18a36f1b4714123706065ed3946a3c3aeb681f98d3e67a3483b088612cb9b0c5322723a0")
    st, pt = B.key_validate(kzg)
    assert st == 0 and B.g1_compress(pt) == kzg
    s = bytes.fromhex(
        "aa0fa03f4dd8cb5a589033651b3b23b384a7b8f5dbd5554f9641d2e60bb4b35b2998fba252137d31bfc02ca5ea09371d"
        "16f3b38e4f1ab19394fcddf61fbe309e4db12cb1c2cca0cac46d25c23c5273f72cfa61b3f270b39655c65837cbaca920")
    st, pt = B.sig_from_bytes(s)
    assert st == 0 and B.g2_in_subgroup(pt) and B.g2_compress(pt) == s


def test_infinity_encodings():
    # crypto/bls.rs:338-343,356-359
    assert B.g2_compress(None) == B.INFINITY_SIGNATURE and B.g1_compress(None) == B.INFINITY_PUBLIC_KEY
    assert B.sig_from_bytes(B.INFINITY_SIGNATURE) == (0, None)
    assert B.key_validate(B.INFINITY_PUBLIC_KEY)[0] == B.BLST_PK_IS_INFINITY
    assert B.key_validate(bytes(48))[0] == B.BLST_BAD_ENCODING  # `zero_public_key` is not decodable
    assert B.eth_fast_aggregate_verify([], b"x" * 32, B.INFINITY_SIGNATURE) == 0
    assert B.fast_aggregate_verify([], b"x" * 32, B.INFINITY_SIGNATURE) == B.BLST_AGGR_TYPE_MISMATCH


def test_pairing_bilinear_nondegenerate():
    e = B.pairing(B.G1, B.G2)
    assert e != B.F12_ONE
    assert B.f12_pow(e, B.R) == B.F12_ONE
    assert B.pairing(B.g1_mul(B.G1, 5), B.g2_mul(B.G2, 7)) == B.f12_pow(e, 35)
    assert B.pairing(B.g1_neg(B.G1), B.G2) == B.f12_conj(e)


def test_final_exponentiation_matches_definition():
    f = B.miller_loop(B.g1_mul(B.G1, 3), B.g2_mul(B.G2, 11))
    assert B.f12_pow(B.final_exponentiation_slow(f), 3) == B.final_exponentiation(f)


def test_psi_and_fast_cofactor_clearing():
    q = B.g2_mul(B.G2, 12345)
    assert B.g2_psi(q) == B.g2_mul(q, B.P % B.R)
    # a point on E2 outside G2: map a field element through SSWU+iso
    u = B.hash_to_field_fp2(b"cofactor", B.DST, 2)[0]
    pt = B.iso3(B.map_to_curve_sswu(u))
    assert B.g2_on_curve(pt) and not B.g2_in_subgroup(pt)
    assert B.clear_cofactor_g2_fast(pt) == B.clear_cofactor_g2(pt)
    assert B.g2_in_subgroup(B.clear_cofactor_g2(pt))


def test_expand_message_xmd_shape():
    out = B.expand_message_xmd(b"abc", B.DST, 256)
    assert len(out) == 256
    dstp = B.DST + bytes([len(B.DST)])
    b0 = hashlib.sha256(bytes(64) + b"abc" + b"\x01\x00" + b"\x00" + dstp).digest()
    assert out[:32] == hashlib.sha256(b0 + b"\x01" + dstp).digest()


def test_aggregate_roundtrip_like_reference_tests():
    # crypto/bls.rs:489-523 with deterministic keys, n reduced for the pure-Python oracle
    n = 4
    sks = [int.from_bytes(hashlib.sha256(b"sk%d" % i).digest(), "big") % B.R for i in range(n)]
    pks = [B.sk_to_pk(s) for s in sks]
    msg = b"message"
    sigs = [B.sign(s, msg) for s in sks]
    st, agg = B.aggregate(sigs)
    assert st == 0
    assert B.fast_aggregate_verify(pks, msg, agg) == 0
    assert B.fast_aggregate_verify(pks[:-1], msg, agg) == B.BLST_VERIFY_FAIL
    msgs = [bytes([i]) * 64 for i in range(n)]
    st, agg2 = B.aggregate([B.sign(s, m) for s, m in zip(sks, msgs)])
    assert B.aggregate_verify(pks, msgs, agg2) == 0
    assert B.aggregate_verify(pks, msgs[::-1], agg2) == B.BLST_VERIFY_FAIL
    assert B.aggregate_verify(pks, msgs[:-1], agg2) == B.BLST_VERIFY_FAIL
    st, apk = B.eth_aggregate_public_keys(pks)
    assert st == 0 and apk == B.sk_to_pk(sum(sks) % B.R)
    assert B.aggregate([])[0] == B.EMPTY_AGGREGATE and B.eth_aggregate_public_keys([])[0] == B.EMPTY_AGGREGATE


def test_edge_verdicts():
    pk = B.sk_to_pk(7)
    sig = B.sign(7, b"m" * 32)
    # flags
    assert B.key_validate(bytes([pk[0] & 0x7F]) + pk[1:])[0] == B.BLST_BAD_ENCODING
    assert B.key_validate(bytes([0xE0]) + bytes(47))[0] == B.BLST_BAD_ENCODING
    assert B.key_validate(bytes([0xC0]) + bytes(46) + b"\x01")[0] == B.BLST_BAD_ENCODING
    # x >= p
    assert B.key_validate(bytes([0x9F]) + b"\xff" * 47)[0] == B.BLST_BAD_ENCODING
    # on curve but not in G1: find small x with a square x^3+4 and r*P != inf
    x = 1
    while True:
        y = B.fp_sqrt((x ** 3 + 4) % B.P)
        if y is not None and not B.g1_in_subgroup((x, y)):
            break
        x += 1
    assert B.key_validate(B.g1_compress((x, y)))[0] == B.BLST_POINT_NOT_IN_GROUP
    # not on curve
    x = 1
    while B.fp_sqrt((x ** 3 + 4) % B.P) is not None:
        x += 1
    assert B.key_validate(bytes([0x80]) + x.to_bytes(48, "big")[1:])[0] == B.BLST_POINT_NOT_ON_CURVE
    # signature not in G2 -> verify fails with POINT_NOT_IN_GROUP (collapsed to InvalidSignature)
    u = B.hash_to_field_fp2(b"edge", B.DST, 2)[0]
    bad = B.g2_compress(B.iso3(B.map_to_curve_sswu(u)))
    assert B.sig_from_bytes(bad)[0] == 0
    assert B.verify_signature(pk, b"m" * 32, bad) == B.VERIFY_POINT_NOT_IN_GROUP
    assert B.error_variant(B.VERIFY_POINT_NOT_IN_GROUP) == "InvalidSignature"   # crypto/bls.rs:72-76
    assert B.error_variant(B.BLST_POINT_NOT_IN_GROUP) == "BLST"                 # a key outside G1: crypto/bls.rs:69
    assert B.aggregate([sig, bad])[0] == B.BLST_POINT_NOT_IN_GROUP              # aggregate maps it to Error::BLST (:92)
    # infinity signature: decodes, never verifies for a valid key
    assert B.verify_signature(pk, b"m" * 32, B.INFINITY_SIGNATURE) == B.BLST_VERIFY_FAIL
    # keys summing to infinity are rejected
    neg = B.g1_compress(B.g1_neg(B.g1_mul(B.G1, 7)))
    assert B.fast_aggregate_verify([pk, neg], b"m" * 32, B.INFINITY_SIGNATURE) == B.VERIFY_PK_IS_INFINITY
    assert B.error_variant(B.VERIFY_PK_IS_INFINITY) == "InvalidSignature" and B.error_variant(B.BLST_PK_IS_INFINITY) == "BLST"
    # error order: first bad key wins over a later one and over a bad signature
    assert B.fast_aggregate_verify([B.INFINITY_PUBLIC_KEY, bytes(48)], b"", bytes(96)) == B.BLST_PK_IS_INFINITY
    assert B.fast_aggregate_verify([pk, bytes(48)], b"", bytes(96)) == B.BLST_BAD_ENCODING
