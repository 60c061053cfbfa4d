"""Seeded lists of public keys / signatures with damaged members for `aggregate` (crypto/bls.rs:79-93) and
`eth_aggregate_public_keys` (:135-148): which member's error wins, infinity members, one-element lists, lists longer than a
wave.  Byte surgery by tests/_blsmutate.py; expectations come from the oracles, not from here.  Test infrastructure."""
import random

from tests import _blsmutate as M


def cases(pks, sigs, n_cases: int, seed: int, max_len: int = 140):
    """pks / sigs: lists of valid 48 / 96-byte encodings (signatures over ONE message).  Yields (kind, members) with kind in
    ("pk", "sig"): a list drawn from the pool with 0..3 members damaged."""
    r = random.Random(seed)
    out = []
    for c in range(n_cases):
        kind = "pk" if c % 2 == 0 else "sig"
        pool = pks if kind == "pk" else sigs
        L = r.choice((1, 1, 2, 3, 5, 8, 13, 31, 64, 65, 70)) if r.random() < 0.8 else r.randrange(1, max_len + 1)
        members = [bytearray(pool[r.randrange(len(pool))]) for _ in range(L)]
        n_bad = r.choice((0, 1, 1, 1, 2, 3))
        for _ in range(min(n_bad, L)):
            pos = r.randrange(L)
            if kind == "pk":
                M.mutate_pk(members[pos], r.randrange(len(M.PK_KINDS)), r, r.randrange(1 << 16))
            else:
                M.mutate_sig(members[pos], r.randrange(len(M.SIG_KINDS)), r, r.randrange(1 << 16))
        out.append((kind, [bytes(m) for m in members]))
    return out


def expect_cpp(kind, members, cbls, cache):
    """(status, bytes | None) by the C++ restatement.  Keys: first failing key in list order (crypto/bls.rs:141), else the sum."""
    if kind == "sig":
        return cbls.aggregate(members)
    for m in members:
        if m not in cache:
            cache[m] = cbls.key_validate(m)
        if cache[m]:
            return cache[m], None
    return cbls.g1_msm(members, [1] * len(members))
