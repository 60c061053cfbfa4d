"""oracle/c (C restatement, used as fast checker and CPU baseline) against oracle/ssz.py."""
import hashlib
import random
import time

import pytest

from oracle import cref, ssz
from tests.test_hostsim_merkle import make_validators


@pytest.mark.parametrize("portable", [0, 1])
def test_c_oracle_matches_python(portable):
    L = cref.lib()
    L.oc_force_portable(portable)
    try:
        r = random.Random(9)
        for n in (0, 1, 55, 56, 64, 119, 1000):
            d = r.randbytes(n)
            assert cref.sha256(d) == hashlib.sha256(d).digest()
        for nbytes, limit in [(0, 8), (32, 1), (100, 16), (32 * 777, 1 << 38), (8 * 1025, 1 << 38)]:
            d = r.randbytes(nbytes)
            got, h = cref.merkleize_bytes(d, limit, 13)
            assert got == ssz.mix_in_length(ssz.merkleize_bytes(d, limit), 13)
            assert h == ssz.hash64_count((nbytes + 31) // 32, limit) + 1
        for n in (0, 1, 5, 33):
            vs = make_validators(n, n)
            ser = b"".join(ssz.Validator.serialize(v) for v in vs)
            got, h = cref.htr_validators(ser)
            assert got == ssz.SSZList(ssz.Validator, 1 << 40).htr(vs)
    finally:
        L.oc_force_portable(0)


def test_c_oracle_speed_smoke():
    from ethereum_consensus_amd import synthetic as S
    v = S.validators(1 << 14).tobytes()
    t = time.time()
    _, h = cref.htr_validators(v)
    dt = time.time() - t
    assert h == 9 * (1 << 14) - 1 + 26 + 1
    print(f"C oracle: {h / dt / 1e6:.2f} M hash64/s single thread (sha-ni={cref.lib().oc_have_shani()})")
