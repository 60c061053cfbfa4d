"""Seeded random mutations of valid (pk, msg, sig) tuples -- the negative space of the decoders and of the status algebra
(crypto/bls.rs:69-70,119-131, 279-285, 330-336): flag bits, x >= p, sign flips, swapped halves of a G2 encoding, curve points
outside the subgroups, infinity encodings with stray bits, all-zero / all-one tails, single-bit damage, wrong messages, and
pairs of faults (which error wins).  Pure byte surgery on the compressed encodings; what each mutant SHOULD return is not
asserted here -- the two oracles (oracle/bls12_381.py, oracle/c/bls12_381.cpp) and the kernels must agree on all of them.

Test infrastructure."""
import random

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB

PK_KINDS = ("pk: compression flag flipped", "pk: infinity flag set on a finite point", "pk: sign flag flipped", "pk: x >= p",
            "pk: one bit of x flipped", "pk: curve point outside G1", "pk: all zero", "pk: 0xff tail", "pk: infinity",
            "pk: infinity with the sign flag", "pk: x = p - 1 - k")
SIG_KINDS = ("sig: compression flag flipped", "sig: infinity flag set on a finite point", "sig: sign flag flipped", "sig: x.c1 >= p",
             "sig: x.c0 >= p", "sig: one bit of x flipped", "sig: curve point outside G2", "sig: halves swapped", "sig: all zero",
             "sig: 0xff tail", "sig: infinity", "sig: infinity with the sign flag", "sig: stray top bits in x.c0")
KINDS = PK_KINDS + SIG_KINDS + ("msg: one bit flipped", "pk and sig both damaged")


def mutate_pk(b: bytearray, k: int, r: random.Random, i: int) -> None:
    from ethereum_consensus_amd import synthetic as syn
    if k == 0:
        b[0] ^= 0x80
    elif k == 1:
        b[0] |= 0x40
    elif k == 2:
        b[0] ^= 0x20
    elif k == 3:
        x = P + r.getrandbits(r.choice((1, 8, 64, 300)))
        flags = b[0] & 0xE0
        b[:] = x.to_bytes(48, "big")
        b[0] = (b[0] & 0x1F) | flags
    elif k == 4:
        bit = r.randrange(3, 384)  # not a flag bit
        b[bit // 8] ^= 0x80 >> (bit % 8)
    elif k == 5:
        b[:] = syn.off_subgroup_public_key(i)
    elif k == 6:
        b[:] = bytes(48)
    elif k == 7:
        b[24:] = b"\xff" * 24
        if r.random() < 0.5:
            b[:] = bytes([b[0] | 0x1F]) + b"\xff" * 47
    elif k == 8:
        b[:] = bytes([0xC0]) + bytes(47)
    elif k == 9:
        b[:] = bytes([0xE0]) + bytes(47)
    elif k == 10:
        x = P - 1 - r.getrandbits(r.choice((0, 4, 32)))
        flags = b[0] & 0xE0
        b[:] = x.to_bytes(48, "big")
        b[0] = (b[0] & 0x1F) | flags
    else:
        raise ValueError(k)


def mutate_sig(b: bytearray, k: int, r: random.Random, i: int) -> None:
    from ethereum_consensus_amd import synthetic as syn
    if k == 0:
        b[0] ^= 0x80
    elif k == 1:
        b[0] |= 0x40
    elif k == 2:
        b[0] ^= 0x20
    elif k == 3:
        x = P + r.getrandbits(r.choice((1, 8, 64, 300)))
        flags = b[0] & 0xE0
        b[:48] = x.to_bytes(48, "big")
        b[0] = (b[0] & 0x1F) | flags
    elif k == 4:
        b[48:] = (P + r.getrandbits(r.choice((1, 8, 64, 300)))).to_bytes(48, "big")
    elif k == 5:
        bit = r.randrange(3, 768)
        if 384 <= bit < 387:
            bit += 3  # the unused top bits of x.c0 have a kind of their own
        b[bit // 8] ^= 0x80 >> (bit % 8)
    elif k == 6:
        b[:] = syn.off_subgroup_signature(i)
    elif k == 7:
        b[:] = bytes(b[48:]) + bytes(b[:48])
    elif k == 8:
        b[:] = bytes(96)
    elif k == 9:
        b[64:] = b"\xff" * 32
        if r.random() < 0.5:
            b[48:] = b"\xff" * 48
    elif k == 10:
        b[:] = bytes([0xC0]) + bytes(95)
    elif k == 11:
        b[:] = bytes([0xE0]) + bytes(95)
    elif k == 12:
        b[48] |= r.choice((0x80, 0x40, 0x20, 0xE0))
    else:
        raise ValueError(k)


def mutate_tuples(pks: bytearray, msgs: bytearray, sigs: bytearray, n: int, every: int = 3, seed: int = 4) -> bytes:
    """damages tuples i = 0 (mod every) in place, kinds drawn uniformly; returns the kind of every tuple (255: untouched)"""
    r = random.Random(seed)
    kind_of = bytearray(b"\xff" * n)
    n_pk, n_sig = len(PK_KINDS), len(SIG_KINDS)
    for i in range(0, n, every):
        k = r.randrange(len(KINDS))
        kind_of[i] = k
        pk = bytearray(pks[48 * i:48 * i + 48])
        sg = bytearray(sigs[96 * i:96 * i + 96])
        if k < n_pk:
            mutate_pk(pk, k, r, i)
        elif k < n_pk + n_sig:
            mutate_sig(sg, k - n_pk, r, i)
        elif k == n_pk + n_sig:
            bit = r.randrange(256)
            msgs[32 * i + bit // 8] ^= 0x80 >> (bit % 8)
        else:
            mutate_pk(pk, r.randrange(n_pk), r, i)
            mutate_sig(sg, r.randrange(n_sig), r, i)
        pks[48 * i:48 * i + 48] = pk
        sigs[96 * i:96 * i + 96] = sg
    return bytes(kind_of)
