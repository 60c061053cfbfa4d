"""GPU parity tests for the BLS12-381 path: the HIP kernels through the C ABI
(ethereum_consensus_amd.bls -> libecgpu.so) against oracle/bls12_381.py on the same inputs, the
reference's own fixed vectors (crypto/bls.rs:530-544, bin/ec/validator/keystores.rs:240-249), and --
at batch sizes the Python oracle cannot reach -- statuses known by construction of the batch."""
import hashlib
import random

import pytest

from oracle import bls12_381 as B
from tests import _blscases as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from ethereum_consensus_amd import _lib, bls
    L = _lib.load(build_if_missing=False)
    assert L.ecgpu_init(-1) == 0, L.ecgpu_last_error()
    return bls


def sk_bytes(sk):
    return sk.to_bytes(32, "big")


def test_reference_kats(gpu):
    # EIP-2335 keystore public key, bin/ec/validator/keystores.rs:240-249
    assert gpu.sk_to_pk_batch(sk_bytes(C.EIP2335_SK)) == C.EIP2335_PK
    # crypto/bls.rs:530-544 test_can_sign: signing on the device reproduces the fixed signature ...
    assert gpu.sign_batch(sk_bytes(C.CAN_SIGN_SK), [C.CAN_SIGN_MSG]) == C.CAN_SIGN_SIG
    pk = gpu.sk_to_pk_batch(sk_bytes(C.CAN_SIGN_SK))
    assert pk == B.sk_to_pk(C.CAN_SIGN_SK)
    # ... and it verifies (bls.rs:543)
    gpu.verify_signature(pk, C.CAN_SIGN_MSG, C.CAN_SIGN_SIG)
    with pytest.raises(gpu.InvalidSignature):
        gpu.verify_signature(pk, C.CAN_SIGN_MSG + b"!", C.CAN_SIGN_SIG)
    gpu.fast_aggregate_verify([pk], C.CAN_SIGN_MSG, C.CAN_SIGN_SIG)
    gpu.aggregate_verify([pk], [C.CAN_SIGN_MSG], C.CAN_SIGN_SIG)


def test_length_checks_like_the_reference(gpu):
    # crypto/bls.rs:372-406,463-487: wrong-length keys/signatures never reach the backend
    with pytest.raises(gpu.InvalidLength):
        gpu.verify_signature(bytes(47), b"m", bytes(96))
    with pytest.raises(gpu.InvalidLength):
        gpu.verify_signature(bytes(48), b"m", bytes(95))
    with pytest.raises(gpu.EmptyAggregate):
        gpu.aggregate([])
    with pytest.raises(gpu.EmptyAggregate):
        gpu.eth_aggregate_public_keys([])


def test_status_algebra_scalar_calls(gpu):
    for pks, msg, sig, eth in C.fav_cases():
        got = gpu.fast_aggregate_verify_status(pks, msg, sig, eth=bool(eth))
        assert got == C.oracle_fav(pks, msg, sig, eth), (len(pks), eth)
    # verify_signature = one key (bls.rs:64-77)
    pks, msg, sig, _ = C.fav_cases()[0]
    assert gpu.verify_signature_status(pks[0], msg, sig) == B.verify_signature(pks[0], msg, sig) == 0


def test_status_algebra_one_batch_variable_k(gpu):
    cases = [c for c in C.fav_cases() if len(c[1]) == 32]
    for eth in (0, 1):
        pk_buf, off, msgs, sigs = b"", [0], b"", b""
        for pks, msg, sig, _ in cases:
            pk_buf += b"".join(pks)
            off.append(off[-1] + len(pks))
            msgs += msg
            sigs += sig
        got = gpu.fast_aggregate_verify_batch(pk_buf, off, msgs, sigs, eth=bool(eth))
        want = bytes(C.oracle_fav(pks, msg, sig, eth) for pks, msg, sig, _ in cases)
        assert got == want


def test_malformed_encodings(gpu):
    r = random.Random(21)
    sk = r.randrange(1, B.R)
    pk, msg = B.sk_to_pk(sk), r.randbytes(32)
    sig = B.sign(sk, msg)
    bad_pks = C.malformed_g1(r)
    bad_sigs = C.malformed_g2(r)
    n = len(bad_pks) + len(bad_sigs)
    pk_buf = b"".join(bad_pks) + pk * len(bad_sigs)
    sig_buf = sig * len(bad_pks) + b"".join(bad_sigs)
    got = gpu.fast_aggregate_verify_batch(pk_buf, None, msg * n, sig_buf)
    want = bytes([B.fast_aggregate_verify([p], msg, sig) for p in bad_pks] +
                 [B.fast_aggregate_verify([pk], msg, s) for s in bad_sigs])
    assert got == want


def test_aggregate_and_eth_aggregate_public_keys(gpu):
    r = random.Random(22)
    sks = [r.randrange(1, B.R) for _ in range(70)]  # > one wave of lanes: exercises the strided sum
    pks = [B.sk_to_pk(s) for s in sks[:9]] + [gpu.sk_to_pk_batch(sk_bytes(s)) for s in sks[9:12]]
    pk_all = gpu.sk_to_pk_batch(b"".join(sk_bytes(s) for s in sks))
    pks70 = [pk_all[48 * i:48 * i + 48] for i in range(70)]
    assert pks70[:9] == pks[:9]
    assert gpu.eth_aggregate_public_keys(pks70) == B.sk_to_pk(sum(sks) % B.R)
    assert gpu.eth_aggregate_public_keys(pks70[:1]) == pks70[0]
    assert gpu.eth_aggregate_public_keys([pks70[0], pks70[0]]) == B.sk_to_pk(2 * sks[0] % B.R)
    st, out = gpu.eth_aggregate_public_keys_status([pks70[0], B.INFINITY_PUBLIC_KEY])
    assert (st, out) == B.eth_aggregate_public_keys([pks70[0], B.INFINITY_PUBLIC_KEY])
    off = B.g1_compress(C.rand_g1_curve_point(r))
    assert gpu.eth_aggregate_public_keys_status([pks70[0], off, bytes(48)])[0] == B.eth_aggregate_public_keys([pks70[0], off, bytes(48)])[0]
    # signatures
    msg = r.randbytes(32)
    sig_all = gpu.sign_batch(b"".join(sk_bytes(s) for s in sks), [msg] * 70)
    sigs = [sig_all[96 * i:96 * i + 96] for i in range(70)]
    assert sigs[0] == B.sign(sks[0], msg)
    agg = gpu.aggregate(sigs)
    assert agg == B.sign(sum(sks) % B.R, msg)
    gpu.fast_aggregate_verify(pks70, msg, agg)
    gpu.eth_fast_aggregate_verify(pks70, msg, agg)
    assert gpu.aggregate([sigs[0], B.INFINITY_SIGNATURE]) == sigs[0]  # infinity is allowed (blst aggregate)
    assert gpu.aggregate([B.INFINITY_SIGNATURE]) == B.INFINITY_SIGNATURE
    off2 = B.g2_compress(C.rand_g2_curve_point(r))
    for lst in ([sigs[0], off2], [off2, bytes(96)], [sigs[0], bytes(96), off2]):
        assert gpu.aggregate_status(lst)[0] == B.aggregate(lst)[0]


def test_aggregate_verify(gpu):
    r = random.Random(23)
    sks = [r.randrange(1, B.R) for _ in range(3)]
    pks = [B.sk_to_pk(s) for s in sks]
    msgs = [r.randbytes(32), r.randbytes(7), b""]
    sig_pts = [B.g2_mul(B.hash_to_g2(m), s) for s, m in zip(sks, msgs)]
    acc = None
    for p in sig_pts:
        acc = B.g2_add(acc, p)
    sig = B.g2_compress(acc)
    cases = [(pks, msgs, sig), (pks, [msgs[1], msgs[0], msgs[2]], sig), (pks[:2], msgs[:2], sig), (pks, msgs[:2], sig),
             ([], [], sig), ([B.INFINITY_PUBLIC_KEY] + pks[1:], msgs, sig), (pks, msgs, bytes(96)),
             (pks, msgs, B.g2_compress(C.rand_g2_curve_point(r))), (pks[:1], msgs[:1], B.g2_compress(sig_pts[0]))]
    for p, m, s in cases:
        assert gpu.aggregate_verify_status(p, m, s) == B.aggregate_verify(p, m, s), (len(p), len(m))


def S(tag, i):
    return hashlib.sha256(b"ecgpu/v1/" + tag + b"/" + i.to_bytes(4, "little")).digest()


def test_batch_4096_with_fault_injection(gpu):
    """SURVEY.md 8(d) config-2 shape at a size the GPU finishes in well under a second: K = 1 tuples
    generated on the device (sk -> pk, sign), every 16th tuple corrupted; the expected status of each
    tuple is known by construction and a 24-tuple sample is cross-checked with the Python oracle."""
    n = 4096
    sks = [1 + int.from_bytes(S(b"sk", i), "big") % (B.R - 1) for i in range(n)]
    skb = b"".join(sk_bytes(s) for s in sks)
    msgs = [S(b"msg", i) for i in range(n)]
    pks = bytearray(gpu.sk_to_pk_batch(skb))
    sigs = bytearray(gpu.sign_batch(skb, msgs))
    msgb = bytearray(b"".join(msgs))
    r = random.Random(31)
    off_g1 = B.g1_compress(C.rand_g1_curve_point(r))
    off_g2 = B.g2_compress(C.rand_g2_curve_point(r))
    want = bytearray(n)
    for i in range(0, n, 16):
        kind = (i // 16) % 8
        if kind == 0:  # wrong message
            msgb[32 * i] ^= 1
            want[i] = B.BLST_VERIFY_FAIL
        elif kind == 1:  # swapped public key
            pks[48 * i:48 * i + 48] = pks[48 * (i + 1):48 * (i + 2)]
            want[i] = B.BLST_VERIFY_FAIL
        elif kind == 2:  # signature outside the G2 subgroup
            sigs[96 * i:96 * i + 96] = off_g2
            want[i] = B.BLST_POINT_NOT_IN_GROUP
        elif kind == 3:  # public key outside the G1 subgroup
            pks[48 * i:48 * i + 48] = off_g1
            want[i] = B.BLST_POINT_NOT_IN_GROUP
        elif kind == 4:  # compression flag cleared
            pks[48 * i] &= 0x7F
            want[i] = B.BLST_BAD_ENCODING
        elif kind == 5:  # x >= p
            sigs[96 * i:96 * i + 48] = bytes([0x9F]) + b"\xff" * 47
            want[i] = B.BLST_BAD_ENCODING
        elif kind == 6:  # pk = infinity
            pks[48 * i:48 * i + 48] = B.INFINITY_PUBLIC_KEY
            want[i] = B.BLST_PK_IS_INFINITY
        else:  # sig = infinity
            sigs[96 * i:96 * i + 96] = B.INFINITY_SIGNATURE
            want[i] = B.BLST_VERIFY_FAIL
    got = gpu.fast_aggregate_verify_batch(bytes(pks), None, bytes(msgb), bytes(sigs))
    assert got == bytes(want)
    for i in list(range(0, 16 * 8, 16)) + r.sample(range(n), 16):
        o = B.fast_aggregate_verify([bytes(pks[48 * i:48 * i + 48])], bytes(msgb[32 * i:32 * i + 32]), bytes(sigs[96 * i:96 * i + 96]))
        assert o == got[i], i
    # linearity: the aggregate of all valid signatures of ONE message verifies against all keys (K = 64)
    m = S(b"att", 0)
    sig64 = gpu.sign_batch(skb[:32 * 64], [m] * 64)
    agg = gpu.aggregate([sig64[96 * j:96 * j + 96] for j in range(64)])
    clean = gpu.sk_to_pk_batch(skb[:32 * 64])
    assert gpu.fast_aggregate_verify_batch(clean[:48 * 63], [0, 63], m, agg) == bytes([B.BLST_VERIFY_FAIL])
    assert gpu.fast_aggregate_verify_batch(clean, [0, 64], m, agg) == b"\x00"
