"""CPU checks of the BLS12-381 lane programs (the ECG_HD code the gfx950 kernels run, compiled by g++
into tests/hostsim) against oracle/bls12_381.py: field tower, curve ops, encodings, subgroup
checks, hash-to-G2, pairing, and the status algebra of the reference wrappers."""
import ctypes
import random

import pytest

from oracle import bls12_381 as B
from tests import _blscases as C
from tests._hostsim import lib

P = B.P


def b48(x):
    return x.to_bytes(48, "big")


def a1(pt):
    return b48(pt[0]) + b48(pt[1]) if pt else bytes(96)


def a2(pt):
    return (b48(pt[0][0]) + b48(pt[0][1]) + b48(pt[1][0]) + b48(pt[1][1])) if pt else bytes(192)


def un1(b):
    return (int.from_bytes(b[:48], "big"), int.from_bytes(b[48:96], "big"))


def un2(b):
    return ((int.from_bytes(b[:48], "big"), int.from_bytes(b[48:96], "big")),
            (int.from_bytes(b[96:144], "big"), int.from_bytes(b[144:192], "big")))


def fp_op(op, a, b=None):
    out = ctypes.create_string_buffer(48)
    rc = lib().hs_fp_op(op, b48(a), b48(b) if b is not None else None, out)
    return rc, int.from_bytes(out.raw, "big")


def fp2_op(op, a, b=None):
    out = ctypes.create_string_buffer(96)
    rc = lib().hs_fp2_op(op, b48(a[0]) + b48(a[1]), (b48(b[0]) + b48(b[1])) if b is not None else None, out)
    return rc, (int.from_bytes(out.raw[:48], "big"), int.from_bytes(out.raw[48:], "big"))


def test_fp_arithmetic_and_lazy_range():
    r = random.Random(1)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 1 << 380]
    vals = edge + [r.randrange(P) for _ in range(200)]
    for _ in range(600):
        a, b = r.choice(vals), r.choice(vals)
        assert fp_op(0, a, b)[1] == a * b % P
        assert fp_op(1, a, b)[1] == (a + b) % P
        assert fp_op(2, a, b)[1] == (a - b) % P
        assert fp_op(3, a)[1] == (-a) % P
        assert fp_op(6, a)[1] == a * a % P
        assert fp_op(8, a)[1] == 2 * a % P
        assert fp_op(7, a)[0] == int(a > (P - 1) // 2)
        assert fp_op(9, a)[0] == int(a == 0)
        assert fp_op(10, a, b)[0] == int(a == b)
    for a in vals + [r.randrange(1 << k) for k in (1, 29, 30, 31, 60, 200, 379) for _ in range(8)]:  # division-steps inversion
        assert fp_op(4, a)[1] == pow(a, P - 2, P)
    for a in vals[:40]:
        rc, v = fp_op(5, a)
        sq = a == 0 or pow(a, (P - 1) // 2, P) == 1
        assert rc == int(sq)
        if sq:
            assert v * v % P == a
    out = ctypes.create_string_buffer(48)
    for _ in range(20):  # long chains keep values in the lazy [0, 2p) representation
        a, b = r.randrange(P), r.randrange(P)
        lib().hs_fp_chain(b48(a), b48(b), 200, out)
        x = a
        for _i in range(200):
            x = (x * b + a - b) % P
        assert int.from_bytes(out.raw, "big") == x


def _limbs(v):
    return [(v >> (30 * i)) & 0x3FFFFFFF if i < 12 else v >> 360 for i in range(13)]


def _sumprod(avals, bvals):
    n = len(avals)
    arr = (ctypes.c_uint32 * (13 * n))
    a = arr(*[w for v in avals for w in _limbs(v)])
    b = arr(*[w for v in bvals for w in _limbs(v)])
    out = (ctypes.c_uint32 * 13)()
    L = lib()
    L.hs_sumprod_raw.restype = ctypes.c_uint64
    ov = L.hs_sumprod_raw(n, a, b, out)
    return ov, sum(int(out[i]) << (30 * i) for i in range(13))


def _row_sumprod(alimbs, blimbs):
    """one sum of products on the row machine (csrc/bls_row.h row_sumprod: limb j of every operand in lane j of a 16-lane row)"""
    n = len(alimbs)
    arr = (ctypes.c_uint32 * (13 * n))
    out = (ctypes.c_uint32 * 16)()
    L = lib()
    L.hs_row_sumprod_raw.restype = ctypes.c_uint64
    ov = L.hs_row_sumprod_raw(n, arr(*[w for v in alimbs for w in v]), arr(*[w for v in blimbs for w in v]), out)
    return ov, [int(x) for x in out]


def test_row_machine_sum_of_products_equals_the_one_lane_sum():
    """row_sumprod (13 iterations of multiply-adds into ONE accumulator per lane, quotient digit from lane 0, window shifted one
    lane down, two carry passes) returns the same integer < 2p as fp_sumprod for 1 .. 7 products -- for normalised operands, for
    lazy ones (limbs == 2^30, the row machine's own output format), for the largest bounds the generator admits -- and no
    lane's 64-bit accumulator ever overflows."""
    r = random.Random(77)
    R = 1 << 390
    Rinv = pow(R, -1, P)

    def lazy(v):  # a representation of v with some limbs pushed to exactly 2^30 (borrowing one from the limb above)
        l = _limbs(v)
        for i in range(12):
            if l[i] == 0 and l[i + 1] > 0 and r.random() < 0.9:
                l[i], l[i + 1] = 1 << 30, l[i + 1] - 1
        return l

    for n in range(1, 8):
        for trial in range(40):
            # sum of bound products <= 600 p^2 (tools/gen_bls_vm3.py LIMIT): bounds up to 2^9 p on one side
            ka = [r.choice([1, 2, 2, 4, 9]) for _ in range(n)]
            kb = [max(1, min(512, 600 // (n * k))) if trial % 3 == 0 else r.choice([1, 2]) for k in ka]
            av = [r.randrange(k * P) if trial % 5 else k * P - 1 - r.randrange(3) for k in ka]
            bv = [r.randrange(k * P) if trial % 7 else k * P - 1 for k in kb]
            if trial == 1:
                av = [sum(0x3FFFFFFF << (30 * i) for i in range(12)) + (1 << 360)] * n  # saturated low limbs
            ov1, want = _sumprod(av, bv)
            assert ov1 == 0
            assert want % P == sum(a * b for a, b in zip(av, bv)) * Rinv % P
            al = [lazy(v) if trial % 2 else _limbs(v) for v in av]
            bl = [lazy(v) if trial % 4 >= 2 else _limbs(v) for v in bv]
            ov, limbs = _row_sumprod(al, bl)
            assert ov == 0, (n, trial)
            assert limbs[13:] == [0, 0, 0]
            assert all(x <= (1 << 30) for x in limbs[:12])
            assert sum(x << (30 * i) for i, x in enumerate(limbs[:13])) == want, (n, trial)


def _rowfield(op, a, b=None):
    """one Fp operation of csrc/bls_rowfield.h on the host's lane vectors: raw limbs in, (rc, limbs) out"""
    arr = ctypes.c_uint32 * 13
    out = (ctypes.c_uint32 * 16)()
    rc = lib().hs_rowfield_op(op, arr(*a), arr(*b) if b is not None else None, out)
    assert rc >= 0, rc
    return rc, [int(x) for x in out]


def test_row_field_operations_against_integers():
    """bls_rowfield.h: Fp with one limb per lane of a 16-lane row -- additions / subtractions / negations (signed limbs resolved in
    two biased passes, comparisons by the sign of the top limb of an exactly carried difference), products, the exponentiation
    chain, canonical forms and zero tests -- on random values, on the edges of [0, 2p] and on lazy limb patterns (limbs == 2^30)."""
    r = random.Random(5)
    R = 1 << 390
    Rinv = pow(R, -1, P)

    def val(l):
        return sum(x << (30 * i) for i, x in enumerate(l[:13]))

    def lazy(v):
        l = _limbs(v)
        for i in range(11):
            if l[i] == 0 and l[i + 1] > 0 and r.random() < 0.9:
                l[i], l[i + 1] = 1 << 30, l[i + 1] - 1
        return l

    edge = [0, 1, 2, P - 1, P, P + 1, 2 * P - 1, 2 * P, (1 << 360) - 1, 1 << 360, (1 << 360) + 1, P - (1 << 330), P + (1 << 30),
            sum(0x3FFFFFFF << (30 * i) for i in range(12)), sum(0x3FFFFFFF << (30 * i) for i in range(12)) + (1 << 360)]
    edge = [v for v in edge if v <= 2 * P]
    vals = edge + [r.randrange(2 * P) for _ in range(25)]
    for a in vals:
        for rep in (_limbs, lazy):
            al = rep(a)
            rc, c = _rowfield(5, al)                           # canon
            assert val(c) == a % P and all(x < (1 << 30) for x in c[:12]) and c[13:] == [0, 0, 0]
            assert _rowfield(8, al)[0] == int(a % P == 0)      # is_zero
            _, n = _rowfield(2, al)                            # neg
            assert val(n) <= 2 * P and val(n) % P == (-a) % P
            _, q = _rowfield(4, al)                            # sqr
            assert val(q) < 2 * P and val(q) % P == a * a * Rinv % P
            for b in r.sample(vals, 6) + [a, (2 * P - a) % (2 * P + 1)]:
                bl = rep(b)
                _, s_ = _rowfield(0, al, bl)
                assert val(s_) < 2 * P + 1 and val(s_) % P == (a + b) % P
                _, d = _rowfield(1, al, bl)
                assert val(d) < 2 * P + 1 and val(d) % P == (a - b) % P
                _, m = _rowfield(3, al, bl)
                assert val(m) < 2 * P and val(m) % P == a * b * Rinv % P
                assert _rowfield(9, al, bl)[0] == int((a - b) % P == 0)
                if a < 2 * P and b < 2 * P:
                    _, sd = _rowfield(6, al, bl)
                    assert val(sd) < 2 * P + 1 and val(sd) % P == (a - 2 * b) % P
    for a in [r.randrange(2 * P) for _ in range(4)] + [1, P - 1]:   # a^((p-3)/4) on Montgomery residues: (aR)^e R^(1-e)
        _, w = _rowfield(7, _limbs(a))
        e = (P - 3) // 4
        assert val(w) % P == pow(a * Rinv % P, e, P) * R % P


def test_row_field_square_roots_signs_and_the_subgroup_check():
    """bls_rowcurve.h: fp2_sqrt / sgn0 / the ZCash sign on a row against integers, and the psi subgroup check of a decoded
    signature (k_sig_group_row) on points inside and outside G2 -- including the real-only and imaginary-only roots."""
    r = random.Random(31)
    L = lib()
    out = ctypes.create_string_buffer(96)
    cases = [(r.randrange(P), r.randrange(P)) for _ in range(12)]
    cases += [(r.randrange(P), 0) for _ in range(4)] + [(0, r.randrange(P)) for _ in range(2)] + [(0, 0), (1, 0), (P - 1, 0), (0, 1)]
    cases += [B.f2_sqr((r.randrange(P), r.randrange(P))) for _ in range(8)]  # squares for sure
    for a in cases:
        rc = L.hs_rowfield_sqrt(b48(a[0]) + b48(a[1]), out)
        assert rc >= 0
        root = (int.from_bytes(out.raw[:48], "big"), int.from_bytes(out.raw[48:], "big"))
        is_sq = B.f2_sqrt(a) is not None
        assert (rc & 1) == int(is_sq), a
        if is_sq:
            assert B.f2_sqr(root) == (a[0] % P, a[1] % P)
        sgn0 = (a[0] & 1) | (int(a[0] == 0) & (a[1] & 1))
        assert (rc >> 1) & 1 == sgn0
        lex = (a[1] > (P - 1) // 2) if a[1] != 0 else (a[0] > (P - 1) // 2)
        assert (rc >> 2) & 1 == int(lex)
    L.hs_g2_in_subgroup_row.restype = ctypes.c_int
    for k in range(4):
        Q = B.g2_mul(B.G2, r.randrange(1, B.R))
        assert L.hs_g2_in_subgroup_row(a2(Q), 0) == 1 and L.hs_g2_in_subgroup_row(a2(Q), 1) == 1
    from tests import _blscases as C
    for k in range(3):
        assert L.hs_g1_in_subgroup_row(a1(B.g1_mul(B.G1, r.randrange(1, B.R)))) == 1
        assert L.hs_g1_in_subgroup_row(a1(C.rand_g1_curve_point(r))) == 0  # on E1, outside G1
    for k in range(4):
        Q = C.rand_g2_curve_point(r)  # on E2, outside G2
        assert L.hs_g2_in_subgroup_row(a2(Q), 0) == 0 and L.hs_g2_in_subgroup_row(a2(Q), 1) == 0


def test_row_decoders_of_signatures_and_keys_against_the_oracle():
    """bls_rowcurve.h r_sig_decode_and_group / r_pk_validate (k_sig_row / k_pk_row: the side stages of a small batch with their
    square roots on rows): status, point and group verdict for valid encodings, points outside the subgroups and every
    malformed class of tests/_blscases.py -- equal to the oracle's Signature::try_from + group check and key_validate."""
    r = random.Random(77)
    L = lib()
    sigs = [B.g2_compress(B.g2_mul(B.G2, r.randrange(1, B.R))) for _ in range(3)] + [B.g2_compress(C.rand_g2_curve_point(r)) for _ in range(2)]
    sigs += [B.INFINITY_SIGNATURE] + C.malformed_g2(r)
    for c in sigs:
        xy = ctypes.create_string_buffer(192)
        inf = ctypes.c_int(0)
        rc = L.hs_sig_row(c, xy, ctypes.byref(inf))
        assert rc >= 0
        wst, wpt = B.g2_decompress(c)
        assert rc & 0xff == wst, c.hex()
        if wst == 0 and wpt is not None:
            assert inf.value == 0 and un2(xy.raw) == wpt
            assert rc >> 8 == (0 if B.g2_in_subgroup(wpt) else 3), c.hex()
        else:
            assert xy.raw == bytes(192) and rc >> 8 == 0 and inf.value == int(wst == 0)
    keys = [B.sk_to_pk(r.randrange(1, B.R)) for _ in range(3)] + [B.g1_compress(C.rand_g1_curve_point(r)) for _ in range(2)]
    keys += [B.INFINITY_PUBLIC_KEY] + C.malformed_g1(r)
    for c in keys:
        xy = ctypes.create_string_buffer(96)
        inf = ctypes.c_int(0)
        st = L.hs_pk_row(c, xy, ctypes.byref(inf))
        wst, wpt = B.key_validate(c)
        assert st == wst, c.hex()
        dst, dpt = B.g1_decompress(c)
        if dst == 0 and dpt is not None:
            assert un1(xy.raw) == dpt and inf.value == 0
        else:
            assert xy.raw == bytes(96) and inf.value == int(dst == 0)


def test_g2_doubling_over_both_row_pairs_of_a_wave():
    """bls_rowcurve.h jac_dbl_quad (k_h2c_finish_quad: one message per wave, the 126 doublings of the cofactor clearing with their six
    products scheduled three deep over the wave's two row pairs) on 64 host lanes: a doubling equals the generic one in both
    pairs -- random points, lazy-free edge coordinates, y = 0, the point at infinity -- and the end of the message stage equals
    the oracle's hash_to_curve."""
    r = random.Random(401)
    L = lib()
    L.hs_g2_dbl_quad.restype = ctypes.c_int

    def jac(pt, z):  # (x z^2, y z^3, z)
        z2 = B.f2_sqr(z)
        return B.f2_mul(pt[0], z2), B.f2_mul(pt[1], B.f2_mul(z2, z)), z

    pts = [jac(C.rand_g2_curve_point(r), (r.randrange(P), r.randrange(P))) for _ in range(6)]
    pts += [jac(B.g2_mul(B.G2, r.randrange(1, B.R)), (1, 0))]
    pts += [((r.randrange(P), r.randrange(P)), (0, 0), (r.randrange(1, P), 5)), ((1, 0), (1, 0), (0, 0)), ((0, 0), (2, 3), (7, 0))]
    for X, Y, Z in pts:
        blob = b"".join(b48(c % P) for c in (X[0], X[1], Y[0], Y[1], Z[0], Z[1]))
        out, ref = ctypes.create_string_buffer(288), ctypes.create_string_buffer(288)
        assert L.hs_g2_dbl_quad(blob, out, ref) == 0
        assert out.raw == ref.raw
    # additions over both pairs: generic pairs, P + P (-> the doubling), P + (-P) (-> infinity), infinity on either side, the
    # same point in two Jacobian representations
    L.hs_g2_add_quad.restype = ctypes.c_int
    blob = lambda J: b"".join(b48(c % P) for c in (J[0][0], J[0][1], J[1][0], J[1][1], J[2][0], J[2][1]))
    A, Bq = C.rand_g2_curve_point(r), C.rand_g2_curve_point(r)
    rz = lambda: (r.randrange(1, P), r.randrange(P))
    inf_pt = ((1, 0), (1, 0), (0, 0))
    pairs = [(jac(A, rz()), jac(Bq, rz())), (jac(A, rz()), jac(A, rz())), (jac(A, (1, 0)), jac(A, (1, 0))), (jac(A, rz()), jac(B.g2_neg(A), rz())),
             (inf_pt, jac(Bq, rz())), (jac(A, rz()), inf_pt), (inf_pt, inf_pt)]
    pairs += [(jac(C.rand_g2_curve_point(r), rz()), jac(C.rand_g2_curve_point(r), rz())) for _ in range(4)]
    for J1, J2 in pairs:
        out, ref = ctypes.create_string_buffer(288), ctypes.create_string_buffer(288)
        assert L.hs_g2_add_quad(blob(J1), blob(J2), out, ref) == 0
        assert out.raw == ref.raw
    for m in (b"", b"abc", bytes(32), r.randbytes(32), r.randbytes(77)):
        xy = ctypes.create_string_buffer(192)
        inf = ctypes.c_int(9)
        L.hs_hash_to_g2_quad(m, len(m), xy, ctypes.byref(inf))
        assert inf.value == 0 and un2(xy.raw) == B.hash_to_g2(m)


def _lin_raw(op, a, b=None):
    arr = ctypes.c_uint32 * 13
    out = arr()
    rc = lib().hs_fp_lin_raw(op, arr(*_limbs(a)), arr(*_limbs(b)) if b is not None else None, out)
    assert rc == 0
    assert all(int(out[i]) < (1 << 30) for i in range(12)), "limbs must come back normalised"
    return sum(int(out[i]) << (30 * i) for i in range(13))


def test_linear_helpers_at_the_edges_of_their_ranges():
    """fp_sub_dbl (X3 = E^2 - 2D of a doubling), fp_gs_lin (3t +- 2z of a cyclotomic squaring) and fp_reduce_below: the right
    residue, inside the promised interval, for operands at both ends of theirs -- the parity tests only ever feed them random
    values."""
    r = random.Random(91)

    def edge(k):  # values around the ends of [0, k p)
        vals = [0, 1, 2, P - 1, P, P + 1, k * P - 1, k * P - 2, (k * P) // 2, (k - 1) * P, (k - 1) * P + 1, (k - 1) * P - 1]
        vals += [r.randrange(k * P) for _ in range(12)]
        vals += [k * P - 1 - r.randrange(1 << 64) for _ in range(4)] + [r.randrange(1 << 64) for _ in range(4)]
        vals += [sum(0x3FFFFFFF << (30 * i) for i in range(12)) % (k * P)]  # saturated low limbs
        return [v for v in vals if 0 <= v < k * P]

    for a in edge(2):
        for b in edge(2):
            res = _lin_raw(0, a, b)
            assert res < 2 * P and res % P == (a - 2 * b) % P
    for op, sign, kt in ((1, 1, 2), (2, -1, 2), (3, 1, 4)):
        for t in edge(kt):
            for z in edge(4):
                res = _lin_raw(op, t, z)
                assert res < 4 * P and res % P == (3 * t + sign * 2 * z) % P
    for op, kin, kout in ((4, 12, 2), (5, 20, 4), (6, 14, 4), (7, 4, 2)):
        for a in edge(kin):
            res = _lin_raw(op, a)
            assert res < kout * P and res % P == a % P


def test_sum_of_products_lazy_bounds_and_column_headroom():
    """fp_sumprod<N>: one Montgomery reduction for N products of lazy operands.  Result < 2p and == sum a b / R whenever
    the sum is below R p (632 p^2); the 64-bit columns never overflow, even for saturated 30-bit limbs."""
    r = random.Random(77)
    RINV = pow(1 << 390, -1, P)
    for n in range(1, 9):
        # operands at the top of the lazy range: n products of (k p - small) values with n k^2 < 632
        k = 1
        while n * (k + 1) ** 2 < 632 and k < 16:
            k += 1
        for _ in range(40):
            av = [k * P - 1 - r.randrange(1 << r.choice([1, 64, 380])) for _ in range(n)]
            bv = [k * P - 1 - r.randrange(1 << r.choice([1, 64, 380])) for _ in range(n)]
            ov, res = _sumprod(av, bv)
            assert ov == 0
            assert res < 2 * P
            assert res % P == sum(x * y for x, y in zip(av, bv)) * RINV % P
        # random reduced operands
        for _ in range(40):
            av = [r.randrange(2 * P) for _ in range(n)]
            bv = [r.randrange(2 * P) for _ in range(n)]
            ov, res = _sumprod(av, bv)
            assert ov == 0 and res < 2 * P
            assert res % P == sum(x * y for x, y in zip(av, bv)) * RINV % P
        # worst-case limb pattern: every low limb saturated, top limb at the 16p ceiling (the value bound is violated
        # on purpose: only the column headroom is under test)
        sat = sum(0x3FFFFFFF << (30 * i) for i in range(12)) + ((16 * P) >> 360 << 360)
        ov, _res = _sumprod([sat] * n, [sat] * n)
        assert ov == 0
    assert lib().hs_column_overflows() == 0


def test_fp2_arithmetic():
    r = random.Random(2)
    v2 = [(0, 0), (1, 0), (0, 1), (P - 1, 0), (0, P - 1), (5, 0), (0, 7)] + [(r.randrange(P), r.randrange(P)) for _ in range(60)]
    for _ in range(200):
        a, b = r.choice(v2), r.choice(v2)
        assert fp2_op(0, a, b)[1] == B.f2_mul(a, b)
        assert fp2_op(1, a)[1] == B.f2_sqr(a)
        assert fp2_op(4, a)[0] == B.f2_sgn0(a)
        assert fp2_op(5, a)[0] == int(B.f2_lex_largest(a))
        assert fp2_op(6, a)[1] == B.f2_mul_xi(a)
        assert fp2_op(7, a, b)[1] == B.f2_add(a, b)
        assert fp2_op(8, a, b)[1] == B.f2_sub(a, b)
        assert fp2_op(9, a)[1] == B.f2_neg(a)
        assert fp2_op(10, a)[1] == B.f2_conj(a)
    for a in v2:
        if a != (0, 0):
            assert fp2_op(2, a)[1] == B.f2_inv(a)
        rc, s = fp2_op(3, a)
        assert rc == int(B.f2_sqrt(a) is not None)
        if rc:
            assert B.f2_sqr(s) == a
        sq = B.f2_sqr(a)
        rc, s = fp2_op(3, sq)
        assert rc == 1 and B.f2_sqr(s) == sq


def test_g1_decode_validate_compress():
    r = random.Random(5)
    L = lib()
    cases = [B.sk_to_pk(r.randrange(1, B.R)) for _ in range(8)]
    cases += [B.g1_compress(C.rand_g1_curve_point(r)) for _ in range(8)]
    cases += C.malformed_g1(r)
    for c in cases:
        xy = ctypes.create_string_buffer(96)
        inf = ctypes.c_int(0)
        st = L.hs_g1_decompress(c, xy, ctypes.byref(inf))
        wst, wpt = B.g1_decompress(c)
        assert st == wst, c.hex()
        if st == 0:
            assert bool(inf.value) == (wpt is None)
            if wpt:
                assert un1(xy.raw) == wpt
        st = L.hs_g1_key_validate(c, xy)
        wst, wpt = B.key_validate(c)
        assert st == wst, c.hex()
        if wpt:
            out = ctypes.create_string_buffer(48)
            L.hs_g1_compress(a1(wpt), 0, out)
            assert out.raw == c
    out = ctypes.create_string_buffer(48)
    L.hs_g1_compress(bytes(96), 1, out)
    assert out.raw == B.INFINITY_PUBLIC_KEY


def test_g1_group_law_special_cases():
    r = random.Random(6)
    L = lib()
    g = B.G1
    pts = [B.g1_mul(g, r.randrange(1, B.R)) for _ in range(6)]
    lists = [pts, [pts[0], pts[0]], [pts[0], B.g1_neg(pts[0])], [pts[0]] * 3, [None, pts[1]], [pts[1], None], [pts[2]],
             [pts[0], B.g1_neg(pts[0]), pts[3]], [pts[0], pts[1], B.g1_neg(B.g1_add(pts[0], pts[1]))]]
    for lst in lists:
        buf = b"".join(a1(p) for p in lst)
        infs = (ctypes.c_int * len(lst))(*[0 if p else 1 for p in lst])
        xy = ctypes.create_string_buffer(96)
        inf = ctypes.c_int(0)
        L.hs_g1_sum(buf, infs, len(lst), xy, ctypes.byref(inf))
        want = None
        for p in lst:
            want = B.g1_add(want, p)
        assert bool(inf.value) == (want is None)
        if want:
            assert un1(xy.raw) == want
    k = r.randrange(B.R)
    xy = ctypes.create_string_buffer(96)
    inf = ctypes.c_int(0)
    L.hs_g1_mul(a1(g), k.to_bytes(32, "big"), xy, ctypes.byref(inf))
    assert un1(xy.raw) == B.g1_mul(g, k)
    # EIP-2335 keystore KAT (bin/ec/validator/keystores.rs:240-249) through the lane programs
    L.hs_g1_mul(a1(g), C.EIP2335_SK.to_bytes(32, "big"), xy, ctypes.byref(inf))
    out = ctypes.create_string_buffer(48)
    L.hs_g1_compress(xy.raw, 0, out)
    assert out.raw == C.EIP2335_PK
    for p in [C.rand_g1_curve_point(r) for _ in range(6)] + pts:
        assert L.hs_g1_in_subgroup(a1(p)) == int(B.g1_in_subgroup(p))


def test_g2_decode_subgroup_compress_and_group_law():
    r = random.Random(7)
    L = lib()
    sigpts = [B.g2_mul(B.G2, r.randrange(1, B.R)) for _ in range(5)]
    off = [C.rand_g2_curve_point(r) for _ in range(5)]
    cases = [B.g2_compress(p) for p in sigpts + off] + C.malformed_g2(r)
    for c in cases:
        xy = ctypes.create_string_buffer(192)
        inf = ctypes.c_int(0)
        st = L.hs_g2_decompress(c, xy, ctypes.byref(inf))
        wst, wpt = B.g2_decompress(c)
        assert st == wst, c.hex()
        if st == 0:
            assert bool(inf.value) == (wpt is None)
            if wpt:
                assert un2(xy.raw) == wpt
                assert L.hs_g2_in_subgroup(a2(wpt)) == int(B.g2_in_subgroup(wpt))
                out = ctypes.create_string_buffer(96)
                L.hs_g2_compress(a2(wpt), 0, out)
                assert out.raw == c
    lists = [sigpts, [sigpts[0], sigpts[0]], [sigpts[0], B.g2_neg(sigpts[0])], [None, sigpts[1]],
             [sigpts[1], None, sigpts[1]], off[:3]]
    for lst in lists:
        buf = b"".join(a2(p) for p in lst)
        infs = (ctypes.c_int * len(lst))(*[0 if p else 1 for p in lst])
        xy = ctypes.create_string_buffer(192)
        inf = ctypes.c_int(0)
        L.hs_g2_sum(buf, infs, len(lst), xy, ctypes.byref(inf))
        want = None
        for p in lst:
            want = B.g2_add(want, p)
        assert bool(inf.value) == (want is None)
        if want:
            assert un2(xy.raw) == want


def test_expand_message_and_hash_to_g2():
    r = random.Random(8)
    L = lib()
    # 9 fixed shapes + 32 random 32-byte messages: both SSWU branches (g(x1) square / not) and both candidates of the
    # single-exponentiation Fp2 root are hit many times over
    for msg in [b"", b"abc", C.CAN_SIGN_MSG, bytes(32), r.randbytes(32), r.randbytes(100), r.randbytes(55),
                r.randbytes(56), r.randbytes(64)] + [r.randbytes(32) for _ in range(32)]:
        out = ctypes.create_string_buffer(256)
        L.hs_xmd(msg, len(msg), out)
        assert out.raw == B.expand_message_xmd(msg, B.DST, 256)
        xy = ctypes.create_string_buffer(192)
        inf = ctypes.c_int(0)
        L.hs_hash_to_g2(msg, len(msg), xy, ctypes.byref(inf))
        assert un2(xy.raw) == B.hash_to_g2(msg)
        xy2 = ctypes.create_string_buffer(192)  # the two-lanes-per-message form of small batches
        L.hs_hash_to_g2_split(msg, len(msg), xy2, ctypes.byref(inf))
        assert xy2.raw == xy.raw
        xy3 = ctypes.create_string_buffer(192)  # ... whose second half runs on a lane PAIR (k_h2c_finish2, bls_g2_pair2.h)
        L.hs_hash_to_g2_pair2(msg, len(msg), xy3, ctypes.byref(inf))
        assert xy3.raw == xy.raw and inf.value == 0
        xy4 = ctypes.create_string_buffer(192)  # ... or on a 16-lane ROW, limb per lane (k_h2c_finish_row, bls_rowcurve.h)
        L.hs_hash_to_g2_row(msg, len(msg), xy4, ctypes.byref(inf), 0)
        assert xy4.raw == xy.raw and inf.value == 0
        xy5 = ctypes.create_string_buffer(192)  # ... with the two SSWU maps on rows as well (k_h2c_map_row)
        L.hs_hash_to_g2_row(msg, len(msg), xy5, ctypes.byref(inf), 1)
        assert xy5.raw == xy.raw and inf.value == 0
        xy6 = ctypes.create_string_buffer(192)  # ... and the end on a row PAIR, one Fp2 component per row (bls_rowpair.h)
        L.hs_hash_to_g2_row(msg, len(msg), xy6, ctypes.byref(inf), 3)
        assert xy6.raw == xy.raw and inf.value == 0
    # crypto/bls.rs:530-544 test_can_sign through the lane programs: [sk] H(msg) compressed
    xy = ctypes.create_string_buffer(192)
    inf = ctypes.c_int(0)
    L.hs_hash_to_g2(C.CAN_SIGN_MSG, len(C.CAN_SIGN_MSG), xy, ctypes.byref(inf))
    out = ctypes.create_string_buffer(192)
    L.hs_g2_mul(xy.raw, C.CAN_SIGN_SK.to_bytes(32, "big"), out, ctypes.byref(inf))
    sig = ctypes.create_string_buffer(96)
    L.hs_g2_compress(out.raw, 0, sig)
    assert sig.raw == C.CAN_SIGN_SIG


def f12_flat(a):
    return b"".join(b48(c) for f6 in a for f2 in f6 for c in f2)


def f12_un(b):
    v = [int.from_bytes(b[48 * i:48 * i + 48], "big") for i in range(12)]
    return (((v[0], v[1]), (v[2], v[3]), (v[4], v[5])), ((v[6], v[7]), (v[8], v[9]), (v[10], v[11])))


_variant_libs = {}


def lib_variant(variant):
    """"" = the sums-of-products build (what every kernel uses on a healthy box); "calls" = the compact-code build of the
    G2 stage kernels for boxes with slow instruction fetch (-DECG_TOWER_CALLS: Fp2 Karatsuba over out-of-line Fp products,
    textbook doubling).  The Fp6 / Fp12 tower and the pairing exist in one form only since round 3."""
    if not variant:
        return lib()
    if variant not in _variant_libs:
        from ethereum_consensus_amd import build
        _variant_libs[variant] = ctypes.CDLL(build.build_hostsim(verbose=False, variant=variant))
    return _variant_libs[variant]


def op12(op, a, b=None, variant=""):
    out = ctypes.create_string_buffer(576)
    lib_variant(variant).hs_fp12_op(op, f12_flat(a), f12_flat(b) if b else None, out)
    return f12_un(out.raw)


@pytest.mark.parametrize("variant", [""])
def test_fp12_tower_and_pairing(variant):
    r = random.Random(11)
    L = lib_variant(variant)
    _op12 = op12
    op12_v = lambda op, a, b=None: _op12(op, a, b, variant)  # noqa: E731

    def rnd12():
        return tuple(tuple((r.randrange(P), r.randrange(P)) for _ in range(3)) for _ in range(2))

    a, b = rnd12(), rnd12()
    assert op12_v(0, a, b) == B.f12_mul(a, b)
    assert op12_v(1, a) == B.f12_sqr(a)
    assert op12_v(2, a) == B.f12_inv(a)
    assert op12_v(3, a) == B.f12_frob(a)
    assert op12_v(4, a) == B.f12_conj(a)
    fe = B.final_exponentiation(a)
    assert op12_v(6, a) == fe
    assert op12_v(5, fe) == B.f12_sqr(fe)  # Granger-Scott squaring on a cyclotomic element
    assert op12_v(7, fe) == B._cyc_pow_x(fe)
    # the exponentiation's long runs of squarings are Karabina-compressed (csrc/bls_tower.h): the identity has z2 = z3 = 0, the
    # one input where the decompression divides by zero unless it is handled
    assert op12_v(7, B.F12_ONE) == B.F12_ONE and op12_v(6, B.F12_ONE) == B.F12_ONE
    for _ in range(3):
        x = B.final_exponentiation(rnd12())
        assert op12_v(7, x) == B._cyc_pow_x(x)
    Pt, Q = B.g1_mul(B.G1, r.randrange(B.R)), B.g2_mul(B.G2, r.randrange(B.R))
    out = ctypes.create_string_buffer(576)
    L.hs_pairing(1, a1(Pt), (ctypes.c_int * 1)(0), a2(Q), (ctypes.c_int * 1)(0), out)
    assert f12_un(out.raw) == B.pairing(Pt, Q)
    P2, Q2 = B.g1_mul(B.G1, r.randrange(B.R)), B.g2_mul(B.G2, r.randrange(B.R))
    L.hs_pairing(2, a1(Pt) + a1(P2), (ctypes.c_int * 2)(0, 0), a2(Q) + a2(Q2), (ctypes.c_int * 2)(0, 0), out)
    assert f12_un(out.raw) == B.final_exponentiation(B.f12_mul(B.miller_loop(Pt, Q), B.miller_loop(P2, Q2)))
    # a pair with a point at infinity contributes 1
    L.hs_pairing(2, a1(Pt) + a1(None), (ctypes.c_int * 2)(0, 1), a2(Q) + a2(Q2), (ctypes.c_int * 2)(0, 0), out)
    assert f12_un(out.raw) == B.pairing(Pt, Q)


@pytest.mark.parametrize("variant", ["", "calls"])
def test_fast_aggregate_verify_status_algebra(variant):
    L = lib_variant(variant)
    for pks, msg, sig, eth in C.fav_cases():
        got = L.hs_fast_aggregate_verify(b"".join(pks), len(pks), msg, len(msg), sig, eth)
        assert got == C.oracle_fav(pks, msg, sig, eth), (len(pks), eth)


@pytest.mark.parametrize("entry", ["hs_vm3_pairing", "hs_row_pairing"])
def test_lane_group_vm_pairing_programs(entry):
    """The generated lane-group programs (tools/gen_bls_vm3.py: Fp registers, sums of products with derived outputs) executed
    with the kernels' lock-step semantics and the kernels' own limb arithmetic: e(P, H) e(-g1, S) after the final
    exponentiation, coefficient by coefficient."""
    r = random.Random(23)
    L = lib()
    fn = getattr(L, entry)
    fn.restype = ctypes.c_int
    out = ctypes.create_string_buffer(576)
    neg_g1 = (B.G1[0], (P - B.G1[1]) % P)
    # a valid (pk, H(m), sig) triple: the product is one
    sk = r.randrange(1, B.R)
    pk = B.g1_mul(B.G1, sk)
    H = B.hash_to_g2(b"vm program check")
    sig = B.g2_mul(H, sk)
    assert fn(a1(pk), a2(H), a2(sig), out) == 1
    assert f12_un(out.raw) == B.F12_ONE
    # unrelated points: equal to the oracle's value, not one
    Pt, Q, S = B.g1_mul(B.G1, r.randrange(B.R)), B.g2_mul(B.G2, r.randrange(B.R)), B.g2_mul(B.G2, r.randrange(B.R))
    assert fn(a1(Pt), a2(Q), a2(S), out) == 0
    want = B.final_exponentiation(B.f12_mul(B.miller_loop(Pt, Q), B.miller_loop(neg_g1, S)))
    got = f12_un(out.raw)
    # the programs compute f^(3 (p^12 - 1)/r) like csrc/bls_pairing.h (the factor 3 keeps == 1 intact)
    assert got == B.f12_mul(B.f12_sqr(want), want) or got == want


def test_no_column_overflow_anywhere_in_the_suite():
    """Every accumulation of the host lane simulator is checked for 64-bit overflow (csrc/bls_fp.h); after everything above
    -- tower, pairing, group laws, hash-to-curve, verification -- the counter must still be zero."""
    L = lib()
    L.hs_column_overflows.restype = ctypes.c_uint64
    # run one more full verification so that the counter covers a complete path even when this test runs alone
    from tests import _blscases as C
    pk = B.sk_to_pk(C.CAN_SIGN_SK)
    assert L.hs_fast_aggregate_verify(bytes(pk), 1, bytes(C.CAN_SIGN_MSG), len(C.CAN_SIGN_MSG), bytes(C.CAN_SIGN_SIG), 0) == 0
    assert L.hs_column_overflows() == 0


def test_two_lane_message_stage_on_the_compact_build():
    """hash_to_g2_map x 2 + hash_to_g2_finish (k_h2c_map_calls / k_h2c_finish_calls on a slow-fetch box) == hash_to_g2 == oracle,
    on the compact-code tower as well"""
    r = random.Random(21)
    L = lib_variant("calls")
    L.hs_hash_to_g2.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    L.hs_hash_to_g2_split.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    for msg in [b"", C.CAN_SIGN_MSG] + [r.randbytes(32) for _ in range(6)]:
        xy, xy2, inf = ctypes.create_string_buffer(192), ctypes.create_string_buffer(192), ctypes.c_int(0)
        L.hs_hash_to_g2(msg, len(msg), xy, ctypes.byref(inf))
        L.hs_hash_to_g2_split(msg, len(msg), xy2, ctypes.byref(inf))
        assert xy.raw == xy2.raw and un2(xy.raw) == B.hash_to_g2(msg)


def test_miller_loop_on_two_lanes_per_tuple_equals_the_one_lane_loop():
    """csrc/bls_pair2.h (k_miller2: lane 2t = the real parts, lane 2t + 1 = the imaginary parts of every Fp2 value of tuple t,
    operands exchanged across the pair): two host threads in lock step run the very lane program, an exchange being a
    rendezvous.  The Miller value equals the one-lane loop's coefficient by coefficient, the final exponentiation of it equals
    the oracle's product of pairings -- incl. a pair with a point at infinity (contributes 1), both pairs at infinity (the
    loop is skipped: 1) and the verification equation e(pk, H) e(-g1, sig) = 1 itself."""
    r = random.Random(31)
    L = lib()
    out, ref = ctypes.create_string_buffer(576), ctypes.create_string_buffer(576)
    zero2 = (ctypes.c_int * 2)(0, 0)
    for trial in range(2):
        P1, Q1 = B.g1_mul(B.G1, r.randrange(B.R)), B.g2_mul(B.G2, r.randrange(B.R))
        P2, Q2 = B.g1_mul(B.G1, r.randrange(B.R)), B.g2_mul(B.G2, r.randrange(B.R))
        L.hs_pairing_split(a1(P1) + a1(P2), zero2, a2(Q1) + a2(Q2), zero2, 1, out)
        L.hs_miller(a1(P1) + a1(P2), zero2, a2(Q1) + a2(Q2), zero2, ref)
        assert out.raw == ref.raw
        L.hs_pairing_split(a1(P1) + a1(P2), zero2, a2(Q1) + a2(Q2), zero2, 0, out)
        assert f12_un(out.raw) == B.final_exponentiation(B.f12_mul(B.miller_loop(P1, Q1), B.miller_loop(P2, Q2)))
    # a pair with a point at infinity contributes 1; both: the loop does not run
    L.hs_pairing_split(a1(P1) + a1(None), (ctypes.c_int * 2)(0, 1), a2(Q1) + a2(Q2), zero2, 0, out)
    assert f12_un(out.raw) == B.pairing(P1, Q1)
    L.hs_pairing_split(a1(P1) + a1(P2), zero2, a2(None) + a2(None), (ctypes.c_int * 2)(1, 1), 0, out)
    assert f12_un(out.raw) == B.F12_ONE
    # the verification equation: sk * g1 against H, -g1 against sk * H
    sk = r.randrange(1, B.R)
    H = B.hash_to_g2(b"two lanes per tuple")
    pk, sig = B.g1_mul(B.G1, sk), B.g2_mul(H, sk)
    L.hs_pairing_split(a1(pk) + a1(B.g1_neg(B.G1)), zero2, a2(H) + a2(sig), zero2, 0, out)
    assert f12_un(out.raw) == B.F12_ONE
    assert L.hs_column_overflows() == 0


def test_final_exponentiation_on_two_lanes_per_tuple_equals_the_one_lane_one():
    """csrc/bls_finalexp2.h (k_finalexp2): the final exponentiation on a lane pair -- easy part, the five exponentiations by x with
    Granger-Scott and compressed squarings, decompression (an Fp2 inversion across the pair), Frobenius maps -- as two host threads
    in lock step.  Its value equals the one-lane routine's coefficient by coefficient and the oracle's; the verdict `is one` is
    the same on both lanes, true for a valid verification equation and false for a forged one."""
    r = random.Random(37)
    L = lib()
    out, ref = ctypes.create_string_buffer(577), ctypes.create_string_buffer(577)
    zero2 = (ctypes.c_int * 2)(0, 0)
    for trial in range(2):
        P1, Q1 = B.g1_mul(B.G1, r.randrange(B.R)), B.g2_mul(B.G2, r.randrange(B.R))
        P2, Q2 = B.g1_mul(B.G1, r.randrange(B.R)), B.g2_mul(B.G2, r.randrange(B.R))
        L.hs_pairing_split(a1(P1) + a1(P2), zero2, a2(Q1) + a2(Q2), zero2, 2, out)
        L.hs_pairing_split(a1(P1) + a1(P2), zero2, a2(Q1) + a2(Q2), zero2, 0, ref)
        assert out.raw[:576] == ref.raw[:576] and out.raw[576] == 0
        assert f12_un(out.raw[:576]) == B.final_exponentiation(B.f12_mul(B.miller_loop(P1, Q1), B.miller_loop(P2, Q2)))
    sk = r.randrange(1, B.R)
    H = B.hash_to_g2(b"two lanes per tuple, to the end")
    pk, sig = B.g1_mul(B.G1, sk), B.g2_mul(H, sk)
    L.hs_pairing_split(a1(pk) + a1(B.g1_neg(B.G1)), zero2, a2(H) + a2(sig), zero2, 2, out)
    assert f12_un(out.raw[:576]) == B.F12_ONE and out.raw[576] == 1
    L.hs_pairing_split(a1(pk) + a1(B.g1_neg(B.G1)), zero2, a2(H) + a2(B.g2_mul(H, sk + 1)), zero2, 2, out)
    assert out.raw[576] == 0
    # both pairs at infinity: the Miller value is 1, and so is its power
    L.hs_pairing_split(a1(P1) + a1(P2), zero2, a2(None) + a2(None), (ctypes.c_int * 2)(1, 1), 2, out)
    assert f12_un(out.raw[:576]) == B.F12_ONE and out.raw[576] == 1
    assert L.hs_column_overflows() == 0
