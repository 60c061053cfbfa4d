#!/usr/bin/env python3
"""Generates ethereum_consensus_amd/csrc/bls_vm3_prog.h: the lane-group programs of the BLS pairing check for the gfx950
"sum-of-products VM" kernels (csrc/bls_vm3.h, bls_vm3.hip) -- round 2's answer to "Fp12 state in LDS, field operations on
LDS-resident operands, a hot loop that fits the instruction cache, >= 2 waves per SIMD, no private-segment traffic".

Machine.  G lanes of a wave own one tuple (one pairing check); its Fp registers (13 x 30-bit limbs each) live in the tuple's
slice of LDS.  A ROUND of class N gives every lane ONE sum of N Fp products with one Montgomery reduction over lazily
reduced operands (csrc/bls_fp.h fp_sumprod<N>: the arithmetic the lane kernels are built from; N = 0: the lane loads a
register instead).  Lanes work in adjacent PAIRS: lane 2j computes the real part and lane 2j+1 the imaginary part of one Fp2
value, each fetches its partner's fresh result through DPP, and each may emit up to three DERIVED registers
c1 * own + c2 * partner + K p (limbs renormalised, no modular correction) next to its result.  That is where every linear
operation of the tower goes: operands of later products are "forms" alpha c0 + beta c1 of an Fp2 value (negations, doublings,
xi-multiples, the (c0 + c1)(c0 - c1) factors of a square, the 3 / 8 / 12-fold multiples of the point-doubling formulas), and
sums that cross values are folded into the sums of products themselves (a +- b c == a * 1 +- b c).  There are no linear
rounds and no modular additions anywhere; the only conditional reductions are the Montgomery reductions of the sums.

Program.  The algorithm of csrc/bls_pairing.h (2-pair Miller loop over |x| on the M-twist with shared squaring; final
exponentiation: easy part, then (x-1)^2 (x+p)(x^2+p^2-1) + 3 with Granger-Scott squarings), restated over sums of Fp2
products with the linear steps folded as above, traced symbolically, list-scheduled into rounds and register-allocated.
The one Fp inversion of the final exponentiation has no parallelism: the trace is cut there (part A .. norm, a
lane-per-tuple inversion kernel, part C).  Every sum is checked against the lazy-reduction bound (sum of bound products
< R / p = 632) at generation time.

Self-contained (no import of oracle/); `--check` simulates the ENCODED programs on random inputs with Python integers and
compares with a direct evaluation of the pairing-check formulas.

    python tools/gen_bls_vm3.py [--lanes 16] > ethereum_consensus_amd/csrc/bls_vm3_prog.h
"""
import argparse
import random
import sys

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
X_ABS = 0xD201000000010000
G1_X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1

LIMIT = 600   # sum over the terms of bound(a) * bound(b), in units of p^2; the Montgomery reduction needs < R / p = 632
MAXN = 7      # products per sum (descriptor: dst + 7 + 7 register numbers)
MAXDER = 4    # derived outputs per lane per round
CONST_BASE = 192  # register numbers >= CONST_BASE name the workgroup's shared constant area, not the tuple's own slice
CLASSES = (3, 4, 7)  # the sums the kernel carries compiled (csrc/bls_vm3.hip): a round of N products runs as the next class up.
                     # Round 2 / 3: (4, 7), ~40 KB of hot loop.  Round 4 adds the 3-term body (+7 KB, still inside the 64 KB
                     # instruction cache): 195 of part A's 407 rounds have at most three terms per sum (the point arithmetic),
                     # -7 % of its multiply slots; 6 -> 7 and 5 -> 7 stay padded (a 6-term body would not fit)


# ------------------------------------------------------------------------------------------------------------------
# integer Fp2 helpers (constants, the checker)
# ------------------------------------------------------------------------------------------------------------------
def i2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def i2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = i2_mul(r, a)
        a = i2_mul(a, a)
        e >>= 1
    return r


# ------------------------------------------------------------------------------------------------------------------
# the trace: Fp nodes, Fp2 pairs, virtual Fp2 values
# ------------------------------------------------------------------------------------------------------------------
class V:
    """a virtual Fp2 value: M applied to the components of a pair; rows of M are the forms (alpha, beta) of c0', c1'"""
    __slots__ = ("pid", "m")

    def __init__(self, pid, m=((1, 0), (0, 1))):
        self.pid = pid
        self.m = m

    def scale(self, c):
        return V(self.pid, tuple(tuple(c * x for x in row) for row in self.m))

    def neg(self):
        return self.scale(-1)

    def conj(self):
        return V(self.pid, (self.m[0], tuple(-x for x in self.m[1])))

    def xi(self):  # (1 + i) * (c0' + c1' i) = (c0' - c1') + (c0' + c1') i
        r0, r1 = self.m
        return V(self.pid, (tuple(a - b for a, b in zip(r0, r1)), tuple(a + b for a, b in zip(r0, r1))))


class Trace:
    def __init__(self):
        self.fp = []     # dicts: kind in/const/res/form, pair, comp | ab, bound, value (const)
        self.pairs = []  # dicts: kind in/const/op, name, c (2 fp ids), forms {(a,b): fp id}, terms [re, im], value (const)
        self.one = self.const("ONE", (1, 0))
        self.inv_operand = None
        self.inv_result = None

    # -- pairs
    def _pair(self, kind, name=None, value=None):
        pid = len(self.pairs)
        c = []
        for comp in (0, 1):
            self.fp.append({"kind": kind if kind != "op" else "res", "pair": pid, "comp": comp, "bound": 1 if kind == "const" else 2,
                            "value": (value[comp] % P) if value is not None else None})
            c.append(len(self.fp) - 1)
        self.pairs.append({"kind": kind, "name": name, "c": c, "forms": {}, "terms": None, "value": value})
        return pid

    def inp(self, name):
        return V(self._pair("in", name))

    def const(self, name, value):
        return V(self._pair("const", name, (value[0] % P, value[1] % P)))

    # -- forms: the Fp node alpha * c0 + beta * c1 of a pair (None when it is identically zero)
    def form(self, pid, ab):
        a, b = ab
        pr = self.pairs[pid]
        if pr["kind"] == "const":
            v = (a * pr["value"][0] + b * pr["value"][1]) % P
            if v == 0:
                return None
            if (a, b) == (1, 0):
                return pr["c"][0]
            if (a, b) == (0, 1):
                return pr["c"][1]
            if ab not in pr["forms"]:
                self.fp.append({"kind": "const", "pair": pid, "ab": ab, "bound": 1, "value": v})
                pr["forms"][ab] = len(self.fp) - 1
            return pr["forms"][ab]
        if (a, b) == (0, 0):
            return None
        if (a, b) == (1, 0):
            return pr["c"][0]
        if (a, b) == (0, 1):
            return pr["c"][1]
        if ab not in pr["forms"]:
            self.fp.append({"kind": "form", "pair": pid, "ab": ab, "bound": 2 * (abs(a) + abs(b)), "value": None})
            pr["forms"][ab] = len(self.fp) - 1
        return pr["forms"][ab]

    def _new_forms(self, pid, abs_):
        """cost of asking pair `pid` for these forms: constants are free (their forms are more constants); a new form of an
        INPUT pair occupies a register for the whole program (inputs live until their last use), so it weighs more than one of
        a short-lived intermediate"""
        pr = self.pairs[pid]
        if pr["kind"] == "const":
            return 0
        w = 8 if pr["kind"] == "in" else 1
        return sum(w for ab in abs_ if ab not in ((0, 0), (1, 0), (0, 1)) and ab not in pr["forms"])

    def row(self, v, r, sign=1):
        return self.form(v.pid, tuple(sign * x for x in v.m[r]))

    # -- a sum of Fp2-level terms -> a new pair
    #    ("mul", X, Y)  X * Y          ("sqr", X)  X^2          ("fpmul", X, k)  (X0 k, X1 k), k an Fp node
    def _fp_terms(self, t):
        re, im = [], []
        if t[0] == "mul":
            _, x, y = t
            # a scalar multiple moves to a constant operand for free (constants' forms are more constants)
            for _ in range(2):
                if self.pairs[y.pid]["kind"] == "const" and x.m[0][1] == 0 and x.m[1][0] == 0 and x.m[0][0] == x.m[1][1] and x.m[0][0] not in (0, 1):
                    c = x.m[0][0]
                    x, y = V(x.pid), y.scale(c)
                x, y = y, x
            # (M X) Y == X (M Y) for the scalar / xi matrices used here (they commute with the Fp2 product): when exactly one
            # side is plain, the matrix may ride on either pair -- keep it off long-lived inputs
            ident = ((1, 0), (0, 1))
            if self.pairs[x.pid]["kind"] == "in" and x.m != ident and y.m == ident and self.pairs[y.pid]["kind"] == "op":
                x, y = V(x.pid), V(y.pid, x.m)
            elif self.pairs[y.pid]["kind"] == "in" and y.m != ident and x.m == ident and self.pairs[x.pid]["kind"] == "op":
                x, y = V(x.pid, y.m), V(y.pid)
            # the minus sign of i^2 goes to whichever side needs fewer new forms
            neg_x1 = tuple(-a for a in x.m[1])
            neg_y1 = tuple(-a for a in y.m[1])
            cost_x = self._new_forms(x.pid, [x.m[0], x.m[1], neg_x1]) + self._new_forms(y.pid, [y.m[0], y.m[1]])
            cost_y = self._new_forms(x.pid, [x.m[0], x.m[1]]) + self._new_forms(y.pid, [y.m[0], y.m[1], neg_y1])
            x0, x1, y0, y1 = self.row(x, 0), self.row(x, 1), self.row(y, 0), self.row(y, 1)
            re.append((x0, y0))
            if cost_x <= cost_y:
                re.append((self.row(x, 1, -1), y1))
            else:
                re.append((x1, self.row(y, 1, -1)))
            im.append((x0, y1))
            im.append((x1, y0))
        elif t[0] == "sqr":  # c X^2 = c (x0 + x1)(x0 - x1) + (2 c x0 x1) i; the factor c rides on the first operand of each product
            x, c = t[1], (t[2] if len(t) > 2 else 1)
            r0, r1 = x.m
            s = tuple(c * (a + b) for a, b in zip(r0, r1))
            d = tuple(a - b for a, b in zip(r0, r1))
            re.append((self.form(x.pid, s), self.form(x.pid, d)))
            im.append((self.form(x.pid, tuple(2 * c * a for a in r0)), self.form(x.pid, r1)))
        elif t[0] == "sqrxi":  # c xi X^2 = c (R - I) + c (R + I) i with R = (x0 + x1)(x0 - x1), I = 2 x0 x1
            x, c = t[1], (t[2] if len(t) > 2 else 1)
            r0, r1 = x.m
            s = self.form(x.pid, tuple(c * (a + b) for a, b in zip(r0, r1)))
            d = self.form(x.pid, tuple(a - b for a, b in zip(r0, r1)))
            x1 = self.form(x.pid, r1)
            re += [(s, d), (self.form(x.pid, tuple(-2 * c * a for a in r0)), x1)]
            im += [(s, d), (self.form(x.pid, tuple(2 * c * a for a in r0)), x1)]
        else:
            _, x, k = t
            re.append((self.row(x, 0), k))
            im.append((self.row(x, 1), k))
        drop = lambda lst: [(a, b) for a, b in lst if a is not None and b is not None]
        return drop(re), drop(im)

    def sum(self, terms, name=None):
        parts = [self._fp_terms(t) for t in terms]
        # split into chunks of at most MAXN products per component; later chunks add the partial sum times ONE
        chunks, cur_re, cur_im = [], [], []
        for re, im in parts:
            if max(len(cur_re) + len(re), len(cur_im) + len(im)) > (MAXN if not chunks else MAXN - 1):
                chunks.append((cur_re, cur_im))
                cur_re, cur_im = [], []
            cur_re += re
            cur_im += im
        chunks.append((cur_re, cur_im))
        acc = None
        for re, im in chunks:
            if acc is not None:
                one = self.pairs[self.one.pid]["c"][0]
                re = re + [(self.pairs[acc]["c"][0], one)]
                im = im + [(self.pairs[acc]["c"][1], one)]
            for lst in (re, im):
                assert len(lst) <= MAXN, len(lst)
                tot = sum(self.fp[a]["bound"] * self.fp[b]["bound"] for a, b in lst)
                assert tot <= LIMIT, ("lazy-reduction bound exceeded", tot, name)
            pid = self._pair("op", name)
            self.pairs[pid]["terms"] = [re, im]
            acc = pid
        return V(acc)

    def mul(self, x, y):
        return self.sum([("mul", x, y)])

    def sqr(self, x):
        return self.sum([("sqr", x)])


T = None  # the trace being built


# ---- Fp12 = sum_k g_k w^k over Fp2, w^6 = xi  (c0 = (g0, g2, g4), c1 = (g1, g3, g5) in the tower of csrc/bls_tower.h) ----
def f12_mul(f, g):
    out = []
    for k in range(6):
        terms = []
        for i in range(6):
            j = (k - i) % 6
            x = f[i].xi() if i + j >= 6 else f[i]
            terms.append(("mul", x, g[j]))
        out.append(T.sum(terms, "f12mul"))
    return out


def f12_sqr(f):
    """f^2 with every cross product taken once (doubled through the operand) and the squares as two products each; absent
    (zero) coefficients of f -- the first steps of the Miller loop -- are skipped"""
    out = []
    for k in range(6):
        terms = []
        for i in range(6):
            j = (k - i) % 6
            if i > j or f[i] is None or f[j] is None:
                continue
            wrap = i + j >= 6
            if i == j:
                terms.append(("sqrxi", f[i]) if wrap else ("sqr", f[i]))
            elif wrap:
                terms.append(("mul", f[j].scale(2).xi(), f[i]))
            else:
                terms.append(("mul", f[i].scale(2), f[j]))
        out.append(T.sum(terms, "f12sqr") if terms else None)
    return out


def f12_mul_by_line(f, l0, l1, l2):
    """f * (l0 + l1 w^2 + l2 w^3): the Miller-loop line shape on the M-twist (csrc/bls_pairing.h)"""
    out = []
    for k in range(6):
        terms = []
        for shift, l in ((0, l0), (2, l1), (3, l2)):
            i = (k - shift) % 6
            terms.append(("mul", l.xi() if i + shift >= 6 else l, f[i]))  # the forms go to the 3 line coefficients, f stays plain
        out.append(T.sum(terms, "line"))
    return out


def f12_conj(f):
    return [f[k].neg() if k & 1 else f[k] for k in range(6)]


def f12_sqr_full(f):
    return f12_sqr(f)


FROB_GAMMA_INT = [i2_pow((1, 1), k * (P - 1) // 6) for k in range(6)]
FROB2_GAMMA_INT = [i2_mul(i2_mul((g[0], -g[1] % P), (1, 0)), g) for g in FROB_GAMMA_INT]  # conj(g) * g: the p^2 Frobenius constants (in Fp)


def f12_frob(f):
    out = []
    for k in range(6):
        xc = f[k].conj()
        out.append(xc if k == 0 else T.mul(xc, T.const(f"FROB{k}", FROB_GAMMA_INT[k])))
    return out


def f12_frob2(f):
    """a -> a^(p^2): g_k -> g_k * gamma_k^(p+1) with gamma_k^(p+1) in Fp"""
    out = []
    for k in range(6):
        g = FROB2_GAMMA_INT[k]
        assert g[1] == 0
        out.append(f[k] if k == 0 else T.mul(f[k], T.const(f"FROB2_{k}", g)))
    return out


def f12_inv(f):
    """1/f for f = A + B w over Fp6 = Fp2[v] (A = (g0, g2, g4), B = (g1, g3, g5)): (A - B w) / (A^2 - v B^2); the Fp6
    inverse by the norm to Fp2, the Fp2 inverse by the norm to Fp -- whose inversion is the cut between part A and part C"""
    a, b = [f[0], f[2], f[4]], [f[1], f[3], f[5]]

    def f6_prod_terms(x, y, k, coef=1):  # coefficient k of x * y in Fp2[v]/(v^3 - xi)
        ts = []
        for i in range(3):
            j = (k - i) % 3
            xx = x[i].scale(coef)
            ts.append(("mul", xx.xi() if i + j >= 3 else xx, y[j]))
        return ts

    # N = A^2 - v B^2: coefficient k of v B^2 is coefficient k-1 of B^2 (times xi when it wraps)
    n = []
    for k in range(3):
        ts = f6_prod_terms(a, a, k)
        kb = (k - 1) % 3
        for i in range(3):
            j = (kb - i) % 3
            xx = b[i].neg()
            wraps = (1 if i + j >= 3 else 0) + (1 if k == 0 else 0)
            for _ in range(wraps):
                xx = xx.xi()
            ts.append(("mul", xx, b[j]))
        n.append(T.sum(ts, "inv_n"))
    c0 = T.sum([("sqr", n[0]), ("mul", n[1].xi().neg(), n[2])], "inv_c0")
    c1 = T.sum([("sqrxi", n[2]), ("mul", n[0].neg(), n[1])], "inv_c1")
    c2 = T.sum([("sqr", n[1]), ("mul", n[0].neg(), n[2])], "inv_c2")
    t = T.sum([("mul", n[0], c0), ("mul", n[2].xi(), c1), ("mul", n[1].xi(), c2)], "inv_t")
    tc = T.pairs[t.pid]["c"]
    d = T._pair("op", "inv_norm")
    T.pairs[d]["terms"] = [[(tc[0], tc[0]), (tc[1], tc[1])], []]
    T.inv_operand = d
    if T.inv_result is None:
        return None
    dinv = T.pairs[T.inv_result.pid]["c"][0]
    ti = T.sum([("fpmul", t.conj(), dinv)], "inv_ti")
    ci = [T.mul(c, ti) for c in (c0, c1, c2)]  # 1 / N
    ra = [T.sum(f6_prod_terms(a, ci, k), "inv_a") for k in range(3)]
    rb = [T.sum(f6_prod_terms(b, ci, k, -1), "inv_b") for k in range(3)]
    return [ra[0], rb[0], ra[1], rb[1], ra[2], rb[2]]


def f12_cyclotomic_sqr(f):
    """Granger-Scott squaring with the 3 t -+ 2 z updates folded into the sums (csrc/bls_tower.h fp12_cyclotomic_sqr_inl):
    with (t0, t1) = fp4_sqr(a, b) = (a^2 + xi b^2, 2 a b):  z0' = 3 t0(z0,z1) - 2 z0, z1' = 3 t1(z0,z1) + 2 z1,
    z4' = 3 t0(z2,z3) - 2 z4, z5' = 3 t1(z2,z3) + 2 z5, z2' = 3 xi t1(z4,z5) + 2 z2, z3' = 3 t0(z4,z5) - 2 z3."""
    z0, z4, z3, z2, z1, z5 = f[0], f[2], f[4], f[1], f[3], f[5]
    one = T.one

    def t0_terms(a, b):  # 3 (a^2 + xi b^2)
        return [("sqr", a, 3), ("sqrxi", b, 3)]

    def t1_terms(a, b, xi=False):  # 3 * 2 a b
        x = a.scale(6)
        return [("mul", x.xi() if xi else x, b)]

    n0 = T.sum(t0_terms(z0, z1) + [("mul", z0.scale(-2), one)], "cyc")
    n1 = T.sum(t1_terms(z0, z1) + [("mul", z1.scale(2), one)], "cyc")
    n4 = T.sum(t0_terms(z2, z3) + [("mul", z4.scale(-2), one)], "cyc")
    n5 = T.sum(t1_terms(z2, z3) + [("mul", z5.scale(2), one)], "cyc")
    n2 = T.sum(t1_terms(z4, z5, xi=True) + [("mul", z2.scale(2), one)], "cyc")
    n3 = T.sum(t0_terms(z4, z5) + [("mul", z3.scale(-2), one)], "cyc")
    return [n0, n2, n4, n1, n3, n5]


def f12_cyc_pow_x(a):
    acc = a
    for b in range(62, -1, -1):
        acc = f12_cyclotomic_sqr(acc)
        if (X_ABS >> b) & 1:
            acc = f12_mul(acc, a)
    return f12_conj(acc)


# ---- Miller loop: point steps as sums of products (csrc/bls_pairing.h formulas, linear steps folded) ----------------------
def miller_dbl_step(Tp, pxy):
    """T = (X, Y, Z) Jacobian on E2, a = 0.  A = X^2, B = Y^2, E = 3A:
         X3 = E^2 - 8 X B,   W = 4 X B - X3 = 12 X B - E^2,   Y3 = E W - 8 B^2,   Z3 = 2 Y Z
         line (scaled):  l0 = E X - 2 B,  l1 = -(E Z^2) xP,  l2 = (Z3 Z^2) yP"""
    X, Y, Z = Tp
    A = T.sqr(X)
    B = T.sqr(Y)
    ZZ = T.sqr(Z)
    YZ = T.mul(Y, Z)
    E = A.scale(3)
    XB = T.mul(X, B)
    X3 = T.sum([("sqr", E), ("mul", XB.scale(-8), T.one)], "dblX3")
    W = T.sum([("mul", XB.scale(12), T.one), ("sqr", E, -1)], "dblW")
    l0 = T.sum([("mul", E, X), ("mul", B.scale(-2), T.one)], "dbll0")
    EZZ = T.mul(E, ZZ)
    Z3 = YZ.scale(2)
    Z3ZZ = T.mul(Z3, ZZ)
    Y3 = T.sum([("mul", E, W), ("sqr", B, -8)], "dblY3")
    pc = T.pairs[pxy.pid]["c"]
    l1 = T.sum([("fpmul", EZZ.neg(), pc[0])], "dbll1")
    l2 = T.sum([("fpmul", Z3ZZ, pc[1])], "dbll2")
    return (X3, Y3, Z3), (l0, l1, l2)


def miller_add_step(Tp, q, pxy):
    """mixed addition T + Q (add-2007-bl shape of csrc/bls_pairing.h miller_add_step_inl):
         U2 = qx Z^2, S2 = qy Z^3, H = U2 - X, r = 2 (S2 - Y), I = 4 H^2, J = H I, V = X I,
         X3 = r^2 - J - 2V, Y3 = r (V - X3) - 2 Y J, Z3 = 2 Z H  (== (Z + H)^2 - Z^2 - H^2)
         line: l0 = r qx - qy Z3, l1 = -r xP, l2 = Z3 yP"""
    X, Y, Z = Tp
    qx, qy = q
    one = T.one
    ZZ = T.sqr(Z)
    H = T.sum([("mul", qx, ZZ), ("mul", X.neg(), one)], "addH")
    qyZ = T.mul(qy, Z)
    rr = T.sum([("mul", qyZ.scale(2), ZZ), ("mul", Y.scale(-2), one)], "addr")
    HH = T.sqr(H)
    I = HH.scale(4)
    J = T.mul(H, I)
    Vv = T.mul(X, I)
    Z3 = T.mul(Z.scale(2), H)
    X3 = T.sum([("sqr", rr), ("mul", J.neg(), one), ("mul", Vv.scale(-2), one)], "addX3")
    VmX3 = T.sum([("mul", Vv, one), ("mul", X3.neg(), one)], "addVmX3")
    Y3 = T.sum([("mul", rr, VmX3), ("mul", Y.scale(-2), J)], "addY3")
    l0 = T.sum([("mul", rr, qx), ("mul", qy.neg(), Z3)], "addl0")
    pc = T.pairs[pxy.pid]["c"]
    l1 = T.sum([("fpmul", rr.neg(), pc[0])], "addl1")
    l2 = T.sum([("fpmul", Z3, pc[1])], "addl2")
    return (X3, Y3, Z3), (l0, l1, l2)


def miller_loop(pairs):
    """pairs: [(pxy, (qx, qy))]; f conjugated (x < 0).  The first iteration's squaring of f = 1 is skipped and its first
    line product is the line itself."""
    Ts = [(q[0], q[1], T.one) for _, q in pairs]
    f = None
    for b in range(62, -1, -1):
        if f is not None:
            f = f12_sqr(f)
        for k, (pxy, q) in enumerate(pairs):
            Ts[k], (l0, l1, l2) = miller_dbl_step(Ts[k], pxy)
            if f is None:
                f = [l0, None, l1, l2, None, None]
            else:
                f = f12_mul_by_line_sparse(f, l0, l1, l2)
        if (X_ABS >> b) & 1:
            for k, (pxy, q) in enumerate(pairs):
                Ts[k], (l0, l1, l2) = miller_add_step(Ts[k], q, pxy)
                f = f12_mul_by_line_sparse(f, l0, l1, l2)
    return f12_conj(f)


def f12_mul_by_line_sparse(f, l0, l1, l2):
    """f12_mul_by_line for an f that may still have absent (zero) coefficients (the first steps of the loop)"""
    if all(x is not None for x in f):
        return f12_mul_by_line(f, l0, l1, l2)
    out = []
    for k in range(6):
        terms = []
        for shift, l in ((0, l0), (2, l1), (3, l2)):
            i = (k - shift) % 6
            if f[i] is None:
                continue
            terms.append(("mul", l.xi() if i + shift >= 6 else l, f[i]))
        out.append(T.sum(terms, "line0") if terms else None)
    return out


def final_exponentiation(f):
    t = f12_mul(f12_conj(f), f12_inv(f))
    t = f12_mul(f12_frob2(t), t)
    a = f12_mul(f12_cyc_pow_x(t), f12_conj(t))
    a = f12_mul(f12_cyc_pow_x(a), f12_conj(a))
    b = f12_mul(f12_cyc_pow_x(a), f12_frob(a))
    c = f12_mul(f12_mul(f12_cyc_pow_x(f12_cyc_pow_x(b)), f12_frob2(b)), f12_conj(b))
    return f12_mul(c, f12_mul(f12_cyclotomic_sqr(t), t))


VERIFY_INPUTS = ["PXY", "HX", "HY", "SX", "SY"]
F12_NAMES = ["F%d" % k for k in range(6)]  # w-power order: c0.c0, c1.c0, c0.c1, c1.c1, c0.c2, c1.c2


def materialize(vs, name):
    """outputs must be plain pairs (M = identity): multiply virtual ones by ONE"""
    out = []
    for v in vs:
        if v.m != ((1, 0), (0, 1)) or T.pairs[v.pid]["kind"] != "op":
            v = T.sum([("mul", v, T.one)], name)
        out.append(v.pid)
    return out


def trace_part_a():
    global T
    T = Trace()
    i = {n: T.inp(n) for n in VERIFY_INPUTS}
    g1n = T.const("G1_NEG", (G1_X, P - G1_Y))
    f = miller_loop([(i["PXY"], (i["HX"], i["HY"])), (g1n, (i["SX"], i["SY"]))])
    fp = materialize(f, "outF")
    f12_inv([V(p) for p in fp])
    return T, fp + [T.inv_operand], [i[n].pid for n in VERIFY_INPUTS]


def trace_part_c():
    global T
    T = Trace()
    fin = [T.inp(n) for n in F12_NAMES]
    T.inv_result = T.inp("DINV")
    e = final_exponentiation(fin)
    return T, materialize(e, "outE"), [v.pid for v in fin] + [T.inv_result.pid]


# ------------------------------------------------------------------------------------------------------------------
# schedule, allocate, encode
# ------------------------------------------------------------------------------------------------------------------
class Program:
    pass


def round_cost(n, nder):
    """issue-cycle model of one round for one wave (measured, profiles/r02a_vm3probe.txt: ~7.2 cycles per multiply-add at
    two waves per SIMD, ~1900 for a round without products)"""
    return (169 * n + 182) * 7.2 + 260 * nder if n else 1900 + 260 * nder


def build_ops(t, outputs):
    """ops: ("sum", pid) for every needed pair, ("derive", pid, [forms]) for forms nobody can emit while producing the
    pair (inputs, or more than 2 * MAXDER forms)"""
    needed_pairs, needed_forms = set(), {}
    stack = list(outputs)
    while stack:
        pid = stack.pop()
        if pid in needed_pairs:
            continue
        needed_pairs.add(pid)
        pr = t.pairs[pid]
        if pr["kind"] != "op":
            continue
        for lst in pr["terms"]:
            for a, b in lst:
                for n in (a, b):
                    nd = t.fp[n]
                    if nd["kind"] == "form":
                        needed_forms.setdefault(nd["pair"], [])
                        if n not in needed_forms[nd["pair"]]:
                            needed_forms[nd["pair"]].append(n)
                    if nd["kind"] in ("res", "form", "in"):
                        stack.append(nd["pair"])
    ops = []          # dicts: kind, pid, n, forms (fp ids), lane_forms [[..],[..]]
    producer = {}     # fp id -> op index
    for pid in sorted(needed_pairs):
        pr = t.pairs[pid]
        forms = needed_forms.get(pid, [])
        if pr["kind"] == "op":
            own = forms[:2 * MAXDER]
            rest = forms[2 * MAXDER:]
            op = {"kind": "sum", "pid": pid, "n": max(len(pr["terms"][0]), len(pr["terms"][1])), "lane_forms": [own[0::2], own[1::2]]}
            ops.append(op)
            for n in pr["c"] + own:
                producer[n] = len(ops) - 1
        else:
            rest = forms
        while rest:
            chunk, rest = rest[:2 * MAXDER], rest[2 * MAXDER:]
            ops.append({"kind": "derive", "pid": pid, "n": 0, "lane_forms": [chunk[0::2], chunk[1::2]]})
            for n in chunk:
                producer[n] = len(ops) - 1
    return ops, producer


def op_reads(t, op):
    pr = t.pairs[op["pid"]]
    if op["kind"] == "derive":
        return list(pr["c"])
    return [n for lst in pr["terms"] for ab in lst for n in ab]


def make_program(t, outputs, inputs, lanes, window):
    ops, producer = build_ops(t, outputs)
    slots = lanes // 2
    nops = len(ops)
    deps = [set() for _ in range(nops)]
    users = [[] for _ in range(nops)]
    for i, op in enumerate(ops):
        for n in op_reads(t, op):
            if n in producer and producer[n] != i:
                deps[i].add(producer[n])
    for i in range(nops):
        for d in deps[i]:
            users[d].append(i)
    cost = [round_cost(op["n"], 0) for op in ops]
    prio = [0.0] * nops
    for i in reversed(range(nops)):
        prio[i] = cost[i] + max((prio[u] for u in users[i]), default=0.0)
    ndeps = [len(d) for d in deps]
    ready = [i for i in range(nops) if ndeps[i] == 0]
    done = [False] * nops
    head = 0
    rounds = []
    while any(not d for d in done):
        while head < nops and done[head]:
            head += 1
        el = [i for i in ready if i < head + window]
        if not el:
            el = list(ready)
        el.sort(key=lambda i: -prio[i])
        # the class of the round: the largest N among the `slots` most urgent ready ops of the leader's kind (sum / derive)
        lead_sum = ops[el[0]]["n"] > 0
        same = [i for i in el if (ops[i]["n"] > 0) == lead_sum]
        ncls = max(ops[i]["n"] for i in same[:slots])
        if ncls:
            ncls = min(c for c in CLASSES if c >= ncls)
        take = [i for i in same if ops[i]["n"] <= ncls][:slots]
        # a cheaper class that still holds the same ops?  (all taken ops smaller than the leader's class)
        ncls = max(ops[i]["n"] for i in take)
        if ncls:
            ncls = min(c for c in CLASSES if c >= ncls)
        rounds.append((ncls, take))
        ts = set(take)
        ready = [i for i in ready if i not in ts]
        for i in take:
            done[i] = True
            for u in users[i]:
                ndeps[u] -= 1
                if ndeps[u] == 0:
                    ready.append(u)
    # ---- register allocation (Fp registers).  0 = TRASH (written by idle lanes, never read); constants and inputs pinned.
    reg = {}
    nxt = 1
    const_nodes = sorted({n for op in ops for n in op_reads(t, op) if t.fp[n]["kind"] == "const"})
    assert len(const_nodes) <= 256 - CONST_BASE
    for k, n in enumerate(const_nodes):  # constants: one copy per workgroup, register numbers CONST_BASE ..
        reg[n] = CONST_BASE + k
    in_nodes = [n for pid in inputs for n in t.pairs[pid]["c"]]
    for n in in_nodes:
        reg[n] = nxt
        nxt += 1
    out_nodes = [n for pid in outputs for n in t.pairs[pid]["c"]]
    last_use = {}
    for r, (_, take) in enumerate(rounds):
        for i in take:
            for n in op_reads(t, ops[i]):
                last_use[n] = r
    pinned = set(const_nodes) | set(out_nodes)
    free, nreg = [], nxt
    release_at = {}
    for n in in_nodes:
        if n not in pinned and n in last_use:
            release_at.setdefault(last_use[n], []).append(reg[n])
    enc_rounds = []
    peak, peak_round, peak_live, born = 0, 0, [], {}
    for r, (ncls, take) in enumerate(rounds):
        # operands are read before anything is written within a round: registers whose last reader is this round are
        # available to this round's results
        free += release_at.pop(r, [])

        def alloc(n):
            nonlocal nreg
            if n not in last_use and n not in pinned:
                return 0  # never read: TRASH
            if free:
                d = free.pop()
            else:
                d = nreg
                nreg += 1
            reg[n] = d
            if n not in pinned:
                release_at.setdefault(last_use[n], []).append(d)
            return d

        row = []
        for i in take:
            op = ops[i]
            pr = t.pairs[op["pid"]]
            for comp in (0, 1):
                if op["kind"] == "sum":
                    dst = alloc(pr["c"][comp])
                    terms = [(reg[a], reg[b]) for a, b in pr["terms"][comp]]
                else:
                    dst = 0
                    terms = [(reg[pr["c"][comp]], 0)]  # N = 0: the lane loads this register
                ders = []
                for n in op["lane_forms"][comp]:
                    a, b = t.fp[n]["ab"]
                    c_own, c_par = (a, b) if comp == 0 else (b, a)
                    k = 2 * (max(0, -a) + max(0, -b))
                    ders.append((alloc(n), c_own, c_par, k))
                row.append((dst, terms, ders))
        enc_rounds.append((ncls, row))
        in_use = nreg - len(free)
        if in_use > peak:
            peak = in_use
            peak_round = r
            peak_live = [n for n in reg if (n in pinned or last_use.get(n, -1) >= r) and (n in const_nodes or n in in_nodes or n in born and born[n] <= r)]
        for i in take:
            pr_ = t.pairs[ops[i]["pid"]]
            for n in (pr_["c"] if ops[i]["kind"] == "sum" else []) + [x for lf in ops[i]["lane_forms"] for x in lf]:
                born[n] = r
    pr = Program()
    pr.rounds = enc_rounds
    pr.nreg = nreg
    pr.lanes = lanes
    pr.const_regs = [(reg[n], t.fp[n]["value"]) for n in const_nodes]
    pr.input_regs = [reg[n] for n in in_nodes]
    pr.output_regs = [reg[n] for n in out_nodes]
    pr.cycles = sum(round_cost(n, max((len(d) for _, _, d in row), default=0)) for n, row in enc_rounds)
    pr.mads = sum((169 * len(terms) + 182) for n, row in enc_rounds if n for _, terms, _ in row if terms)
    pr.slot_mads = sum((169 * n + 182) * lanes for n, row in enc_rounds if n)
    pr.hist = {}
    for n, row in enc_rounds:
        pr.hist[n] = pr.hist.get(n, 0) + 1
    pr.nops = nops
    pr.nder = sum(len(d) for _, row in enc_rounds for _, _, d in row)
    pr.peak_round = peak_round
    pr.peak_live = [((t.pairs[t.fp[n]["pair"]]["name"] or t.pairs[t.fp[n]["pair"]]["kind"]), t.fp[n]["kind"]) for n in peak_live]
    return pr


# descriptor of one lane in one round: 8 dwords
#   w0: dst | a0 << 8 | a1 << 16 | a2 << 24     w1: a3 | a4 << 8 | a5 << 16 | a6 << 24
#   w2: b0 | b1 << 8 | b2 << 16 | b3 << 24      w3: b4 | b5 << 8 | b6 << 16
#   w4 + d (d < 4): derived output d: reg | (c_own & 255) << 8 | (c_partner & 255) << 16 | K << 24      (reg 0 = none)
# round header: N | nder << 8 (nder = the largest number of derived outputs of any lane of the round)
def encode(pr):
    assert pr.nreg <= CONST_BASE, pr.nreg
    words, hdr = [], []
    for n, row in pr.rounds:
        nder = max((len(d) for _, _, d in row), default=0)
        hdr.append(n | nder << 8)
        for k in range(pr.lanes):
            w = [0] * 8
            if k < len(row):
                dst, terms, ders = row[k]
                regs = [dst] + [a for a, _ in terms] + [0] * (7 - len(terms))
                bs = [b for _, b in terms] + [0] * (7 - len(terms))
                w[0] = regs[0] | regs[1] << 8 | regs[2] << 16 | regs[3] << 24
                w[1] = regs[4] | regs[5] << 8 | regs[6] << 16 | regs[7] << 24
                w[2] = bs[0] | bs[1] << 8 | bs[2] << 16 | bs[3] << 24
                w[3] = bs[4] | bs[5] << 8 | bs[6] << 16
                for d, (r, co, cp, kk) in enumerate(ders):
                    assert -128 <= co <= 127 and -128 <= cp <= 127 and 0 <= kk <= 255
                    w[4 + d] = r | (co & 255) << 8 | (cp & 255) << 16 | kk << 24
            words += w
    return words, hdr


def simulate(pr, words, hdr, inputs):
    """the ENCODED program on Python integers mod p, with the lock-step semantics of the kernel (all reads of a round before
    its writes; register 0 is TRASH and reads as 0 -- unused operand slots multiply TRASH by TRASH)"""
    R = [0] * 256
    for r, v in pr.const_regs:
        R[r] = v
    for r, v in zip(pr.input_regs, inputs):
        R[r] = v
    L = pr.lanes
    sgn = lambda x: x - 256 if x >= 128 else x
    for rd in range(len(hdr)):
        n = hdr[rd] & 255
        own = []
        for k in range(L):
            w = words[(rd * L + k) * 8:(rd * L + k) * 8 + 8]
            regs = [w[0] & 255, w[0] >> 8 & 255, w[0] >> 16 & 255, w[0] >> 24, w[1] & 255, w[1] >> 8 & 255, w[1] >> 16 & 255, w[1] >> 24]
            bs = [w[2] & 255, w[2] >> 8 & 255, w[2] >> 16 & 255, w[2] >> 24, w[3] & 255, w[3] >> 8 & 255, w[3] >> 16 & 255]
            if n == 0:
                own.append(R[regs[1]])
            else:
                own.append(sum(R[regs[1 + j]] * R[bs[j]] for j in range(n)) % P)
        writes = []
        for k in range(L):
            w = words[(rd * L + k) * 8:(rd * L + k) * 8 + 8]
            if n:
                writes.append((w[0] & 255, own[k]))
            for d in range(4):
                x = w[4 + d]
                if x & 255:
                    writes.append((x & 255, (sgn(x >> 8 & 255) * own[k] + sgn(x >> 16 & 255) * own[k ^ 1]) % P))
        for r, v in writes:
            if r:
                R[r] = v
    return [R[r] for r in pr.output_regs]


# ------------------------------------------------------------------------------------------------------------------
# independent evaluation of the pairing check with Python integers (for --check): the textbook tower
# ------------------------------------------------------------------------------------------------------------------
def _ref_f12_mul(a, b):
    out = []
    for k in range(6):
        acc = (0, 0)
        for i in range(6):
            j = (k - i) % 6
            t = i2_mul(a[i], b[j])
            if i + j >= 6:
                t = i2_mul(t, (1, 1))
            acc = ((acc[0] + t[0]) % P, (acc[1] + t[1]) % P)
        out.append(acc)
    return out


def _ref_f12_pow(a, e):
    r = [(1, 0)] + [(0, 0)] * 5
    while e:
        if e & 1:
            r = _ref_f12_mul(r, a)
        a = _ref_f12_mul(a, a)
        e >>= 1
    return r


def _ref_miller(px, py, qx, qy):
    """affine Miller loop on the M-twist, lines scaled by w^3 (as in oracle/bls12_381.py, restated here so that the checker
    stays self-contained)"""
    inv = lambda a: i2_pow(a, P * P - 2)
    sub = lambda a, b: ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
    f = [(1, 0)] + [(0, 0)] * 5
    Tx, Ty = qx, qy

    def line(lam, xT, yT):
        c0 = sub(i2_mul(lam, xT), yT)
        c2 = ((-lam[0] * px) % P, (-lam[1] * px) % P)
        return [c0, (0, 0), c2, (py % P, 0), (0, 0), (0, 0)]

    for bit in bin(X_ABS)[3:]:
        lam = i2_mul(i2_mul(i2_mul(Tx, Tx), (3, 0)), inv(i2_mul(Ty, (2, 0))))
        f = _ref_f12_mul(_ref_f12_mul(f, f), line(lam, Tx, Ty))
        x3 = sub(sub(i2_mul(lam, lam), Tx), Tx)
        Ty = sub(i2_mul(lam, sub(Tx, x3)), Ty)
        Tx = x3
        if bit == "1":
            lam = i2_mul(sub(qy, Ty), inv(sub(qx, Tx)))
            f = _ref_f12_mul(f, line(lam, Tx, Ty))
            x3 = sub(sub(i2_mul(lam, lam), Tx), qx)
            Ty = sub(i2_mul(lam, sub(Tx, x3)), Ty)
            Tx = x3
    return [f[k] if k % 2 == 0 else ((-f[k][0]) % P, (-f[k][1]) % P) for k in range(6)]


R_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def mont_limbs(v):
    m = v % P * (1 << 390) % P
    return [(m >> (30 * i)) & 0x3FFFFFFF for i in range(13)]


def emit(pa, pc, wa, ha, wc, hc):
    def arr(name, vals, per=8, ty="unsigned int"):
        out = [f"static const {ty} {name}[{max(1, len(vals))}] = {{"]
        for i in range(0, len(vals), per):
            out.append("    " + ", ".join("0x%08xu" % v for v in vals[i:i + per]) + ",")
        out.append("};")
        return "\n".join(out)

    print("// GENERATED by tools/gen_bls_vm3.py -- do not edit.  Sum-of-products lane-group programs of the BLS pairing check.")
    print("// One round = one header word (N | nder << 8) + ECG_VM3_<part>_LANES descriptors of 8 dwords (tools/gen_bls_vm3.py encode()).")
    print("#pragma once")
    for tag, pr, w, h in (("A", pa, wa, ha), ("C", pc, wc, hc)):
        fill = pr.mads / max(1, pr.slot_mads)
        print(f"// part {tag}: {pr.nops} ops, {pr.nder} derived outputs in {len(pr.rounds)} rounds {dict(sorted(pr.hist.items()))}, {pr.nreg} registers, "
              f"{pr.mads} multiply-adds per tuple at {100 * fill:.0f} % slot fill, model {int(pr.cycles)} cycles per wave")
        print(f"#define ECG_VM3_{tag}_LANES {pr.lanes}")
        print(f"#define ECG_VM3_{tag}_NREG {pr.nreg}")
        print(f"#define ECG_VM3_CONST_BASE {CONST_BASE}") if tag == "A" else None
        print(f"#define ECG_VM3_{tag}_ROUNDS {len(pr.rounds)}")
        print(f"#define ECG_VM3_{tag}_NIN {len(pr.input_regs)}")
        print(f"#define ECG_VM3_{tag}_NOUT {len(pr.output_regs)}")
        print(f"#define ECG_VM3_{tag}_MADS {pr.mads}")
        print(arr(f"ECG_VM3_{tag}_IN", pr.input_regs))
        print(arr(f"ECG_VM3_{tag}_OUT", pr.output_regs))
        print(f"#define ECG_VM3_{tag}_NCONST {len(pr.const_regs)}")
        print(arr(f"ECG_VM3_{tag}_CONST_REG", [r for r, _ in pr.const_regs]))
        print("// constant values: 13 x 30-bit limbs each, Montgomery form (R = 2^390), one row per constant")
        print(arr(f"ECG_VM3_{tag}_CONST_VAL", [x for _, v in pr.const_regs for x in mont_limbs(v)], per=13))
        print(arr(f"ECG_VM3_{tag}_HDR", h, per=16))
        print(arr(f"ECG_VM3_{tag}_PROG", w))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=16, help="lanes per tuple in part A (the Miller loops)")
    ap.add_argument("--lanes-c", type=int, default=0, help="lanes per tuple in part C (the final exponentiation); 0 = as part A")
    ap.add_argument("--window", type=int, default=400)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--stats", action="store_true")
    args = ap.parse_args()
    sys.setrecursionlimit(100000)
    ta, outs_a, ins_a = trace_part_a()
    pa = make_program(ta, outs_a, ins_a, args.lanes, args.window)
    tc, outs_c, ins_c = trace_part_c()
    pc = make_program(tc, outs_c, ins_c, args.lanes_c or args.lanes, args.window)
    wa, ha = encode(pa)
    wc, hc = encode(pc)
    if args.stats or args.check:
        for tag, pr in (("A", pa), ("C", pc)):
            sys.stderr.write(f"part {tag}: {pr.nops} ops, {pr.nder} derived, rounds {len(pr.rounds)} {dict(sorted(pr.hist.items()))}, nreg {pr.nreg}, "
                             f"mads/tuple {pr.mads}, slot fill {100 * pr.mads / max(1, pr.slot_mads):.0f} %, model {int(pr.cycles)} cycles\n")
        tot = pa.cycles / (64 // pa.lanes) + pc.cycles / (64 // pc.lanes)
        sys.stderr.write(f"model: {int(tot)} cycles of SIMD time per tuple -> {tot * 65536 / 2048 / 2.4e9 * 1e3 * 2:.1f} ms per 65536 tuples "
                         f"(1024 SIMDs, two waves each taking the modelled cycles of SIMD time, 2.4 GHz)\n")
    if args.check:
        rnd = random.Random(1)
        for trial in range(2):
            # a VALID check (e(a G1, b G2) e(-G1, ab G2) == 1) needs curve arithmetic; random field elements exercise the
            # formulas just as well: compare f after the final exponentiation with the textbook evaluation
            env = {nm: (rnd.randrange(P), rnd.randrange(P)) for nm in VERIFY_INPUTS}
            flat = [c for nm in VERIFY_INPUTS for c in env[nm]]
            mid = simulate(pa, wa, ha, flat)
            f_vm = [(mid[2 * k], mid[2 * k + 1]) for k in range(6)]
            fa = _ref_miller(env["PXY"][0], env["PXY"][1], env["HX"], env["HY"])
            fb = _ref_miller(G1_X, P - G1_Y, env["SX"], env["SY"])
            want = _ref_f12_pow(_ref_f12_mul(fa, fb), 3 * (P**12 - 1) // R_ORDER)
            d = mid[12]
            dinv = pow(d, P - 2, P)
            got = simulate(pc, wc, hc, [c for k in range(6) for c in f_vm[k]] + [dinv, 0])
            got = [(got[2 * k], got[2 * k + 1]) for k in range(6)]
            assert got == want, "encoded programs disagree with the textbook pairing value"
        sys.stderr.write("check ok\n")
        return
    if not args.stats:
        emit(pa, pc, wa, ha, wc, hc)


if __name__ == "__main__":
    main()


def pressure_report(t, outputs, inputs, lanes, window, top=12):
    """diagnostic: which values are live at the round of peak register pressure"""
    ops, producer = build_ops(t, outputs)
    # re-run the scheduler part of make_program to get rounds (kept in sync by calling it and reading back)
    pr = make_program(t, outputs, inputs, lanes, window)
    return pr
