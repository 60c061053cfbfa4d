#!/usr/bin/env python3
"""Generates ethereum_consensus_amd/csrc/bls_vm2_prog.h: the lane-group programs of the BLS pairing
check for the gfx950 "Fp2 VM" kernels (csrc/bls_vm2.h, bls_vm2.hip).

Why (measured, profiles/r01b_bls_occupancy_probe.txt, r01c): one lane per pairing keeps every Fp12
temporary in the private segment and is bound by scratch traffic to HBM (92 ms / 65 536 checks);
the first lane-group VM (Fp registers, one Fp operation per lane per round, removed after r01e)
removed the scratch but paid one LDS round trip + barrier per ~50-instruction addition round
(110 ms).  This generator raises the unit of work to Fp2:

  * a register is an Fp2 (26 dwords) in the tuple's slice of LDS;
  * a PRODUCT round gives every lane of the group one Fp2 product / square / scaling
    (3 or 2 Montgomery products, Karatsuba additions done in VGPRs, ~6 700 issue cycles), so the
    LDS latency of a round is a few % of its arithmetic and one wave per SIMD is enough;
  * a LINEAR round gives every lane one Fp2 addition / subtraction / conjugate / xi-twist.

The program is the algorithm of csrc/bls_pairing.h (2-pair Miller loop over |x| with shared
squaring on the M-twist, final exponentiation with Granger-Scott squarings), traced symbolically
over Fp2 values, list-scheduled into rounds of at most G operations and register-allocated.  The one
Fp inversion of the final exponentiation has no parallelism: the trace is cut there (part A ..
norm, lane-per-tuple inversion kernel, part C).

Self-contained (no import of oracle/); `--check` simulates the ENCODED programs on random inputs
with Python integers and compares with a direct evaluation of the traced expressions.

    python tools/gen_bls_vm2.py [--lanes 16] > ethereum_consensus_amd/csrc/bls_vm2_prog.h
"""
import argparse
import random
import sys

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
X_ABS = 0xD201000000010000

# instruction word: op[31:28] flags[27:24] dst[23:16] a[15:8] b[7:0]
OP_NOP, OP_MUL, OP_SQR, OP_MULFP, OP_NORM, OP_LIN, OP_LINXI = range(7)
# flags: LIN   bit0 negate b.c0, bit1 negate b.c1          d = a + (+-b0, +-b1)
#        LINXI bit0 subtract                                 d = a +- xi*b
#        MULFP bit0 use b.c1 (else b.c0) as the Fp scalar    d = (a0 k, a1 k)
#        NORM                                                d = (a0^2 + a1^2, 0)
# round class bits (one byte per round, wave-uniform): which variants are present
CLS_MUL, CLS_SQR, CLS_MULFP, CLS_NORM, CLS_LIN, CLS_LINXI = 1, 2, 4, 8, 16, 32
CLS_OF = {OP_MUL: CLS_MUL, OP_SQR: CLS_SQR, OP_MULFP: CLS_MULFP, OP_NORM: CLS_NORM, OP_LIN: CLS_LIN, OP_LINXI: CLS_LINXI}
PROD_OPS = (OP_MUL, OP_SQR, OP_MULFP, OP_NORM)

K_IN, K_CONST, K_OP = "in", "const", "op"


# ------------------------------------------------------------------------------------------------
# Fp2 integer arithmetic (for constants, the checker)
# ------------------------------------------------------------------------------------------------
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


def i2_eval(op, fl, a, b):
    if op == OP_MUL:
        return i2_mul(a, b)
    if op == OP_SQR:
        return i2_mul(a, a)
    if op == OP_MULFP:
        k = b[1] if fl & 1 else b[0]
        return (a[0] * k % P, a[1] * k % P)
    if op == OP_NORM:
        return ((a[0] * a[0] + a[1] * a[1]) % P, 0)
    if op == OP_LIN:
        return ((a[0] + (-b[0] if fl & 1 else b[0])) % P, (a[1] + (-b[1] if fl & 2 else b[1])) % P)
    if op == OP_LINXI:
        x = ((b[0] - b[1]) % P, (b[0] + b[1]) % P)
        return ((a[0] - x[0]) % P, (a[1] - x[1]) % P) if fl & 1 else ((a[0] + x[0]) % P, (a[1] + x[1]) % P)
    raise ValueError(op)


# ------------------------------------------------------------------------------------------------
# expression graph over Fp2
# ------------------------------------------------------------------------------------------------
class Graph:
    def __init__(self):
        self.nodes = []  # (kind, op, flags, a, b, name)
        self.memo = {}
        self.const_values = {}
        self.inv_operand = None   # Fp2 node whose c0 is the Fp element to invert (part A)
        self.inv_result = None    # input node carrying the inverse in c0 (part C)
        self.zero = self.const("ZERO", (0, 0))
        self.one = self.const("ONE", (1, 0))

    def _new(self, key):
        if key in self.memo:
            return self.memo[key]
        self.nodes.append(key)
        self.memo[key] = len(self.nodes) - 1
        return len(self.nodes) - 1

    def inp(self, name):
        return self._new((K_IN, 0, 0, None, None, name))

    def const(self, name, value):
        self.const_values[name] = (value[0] % P, value[1] % P)
        return self._new((K_CONST, 0, 0, None, None, name))

    def op(self, op, fl, a, b):
        return self._new((K_OP, op, fl, a, b, None))

    # -- products
    def mul(self, a, b):
        if a == self.zero or b == self.zero:
            return self.zero
        if a == self.one:
            return b
        if b == self.one:
            return a
        if a == b:
            return self.sqr(a)
        if a > b:
            a, b = b, a
        return self.op(OP_MUL, 0, a, b)

    def sqr(self, a):
        if a in (self.zero, self.one):
            return a
        return self.op(OP_SQR, 0, a, a)

    def mulfp(self, a, k, sel):
        """(a0 k, a1 k) with k = component `sel` of register k"""
        if a == self.zero:
            return self.zero
        return self.op(OP_MULFP, sel, a, k)

    def norm(self, a):
        return self.op(OP_NORM, 0, a, a)

    # -- linear
    def lin(self, a, b, fl):
        if b == self.zero:
            return a
        return self.op(OP_LIN, fl, a, b)

    def add(self, a, b):
        if a == self.zero:
            return b
        if b == self.zero:
            return a
        if a > b:
            a, b = b, a
        return self.op(OP_LIN, 0, a, b)

    def sub(self, a, b):
        if b == self.zero:
            return a
        if a == b:
            return self.zero
        return self.op(OP_LIN, 3, a, b)

    def neg(self, a):
        return self.sub(self.zero, a)

    def conj(self, a):
        if a in (self.zero, self.one):
            return a
        return self.op(OP_LIN, 2, self.zero, a)

    def addxi(self, a, b):
        """a + xi b"""
        if b == self.zero:
            return a
        return self.op(OP_LINXI, 0, a, b)

    def subxi(self, a, b):
        if b == self.zero:
            return a
        return self.op(OP_LINXI, 1, a, b)


G = None  # the graph being traced


# ---- Fp2 layer: thin names over the graph ----------------------------------------------------------
def f2_add(a, b):
    return G.add(a, b)


def f2_sub(a, b):
    return G.sub(a, b)


def f2_neg(a):
    return G.neg(a)


def f2_dbl(a):
    return G.add(a, a)


def f2_conj(a):
    return G.conj(a)


def f2_mul_xi(a):
    return G.addxi(G.zero, a)


def f2_mul(a, b):
    return G.mul(a, b)


def f2_sqr(a):
    return G.sqr(a)


def f2_mul3(a):
    return f2_add(f2_dbl(a), a)


def f2_inv(a):
    """1/a = conj(a) / norm(a); the Fp inversion is the cut between part A and part C"""
    G.inv_operand = G.norm(a)
    if G.inv_result is None:
        return G.zero  # part A only wants the operand
    return f2_conj(G.mulfp(a, G.inv_result, 0))


# ---- Fp6 -------------------------------------------------------------------------------------------
def f6_add(a, b):
    return tuple(f2_add(x, y) for x, y in zip(a, b))


def f6_sub(a, b):
    return tuple(f2_sub(x, y) for x, y in zip(a, b))


def f6_neg(a):
    return tuple(f2_neg(x) for x in a)


def f6_mul_v(a):
    return (f2_mul_xi(a[2]), a[0], a[1])


def f6_mul(a, b):
    t0 = f2_mul(a[0], b[0])
    t1 = f2_mul(a[1], b[1])
    t2 = f2_mul(a[2], b[2])
    m12 = f2_mul(f2_add(a[1], a[2]), f2_add(b[1], b[2]))
    m01 = f2_mul(f2_add(a[0], a[1]), f2_add(b[0], b[1]))
    m02 = f2_mul(f2_add(a[0], a[2]), f2_add(b[0], b[2]))
    c0 = G.addxi(t0, f2_sub(f2_sub(m12, t1), t2))
    c1 = G.addxi(f2_sub(f2_sub(m01, t0), t1), t2)
    c2 = f2_add(f2_sub(f2_sub(m02, t0), t2), t1)
    return (c0, c1, c2)


def f6_mul_by_01(a, c0, c1):
    t0 = f2_mul(a[0], c0)
    t1 = f2_mul(a[1], c1)
    mid = f2_sub(f2_sub(f2_mul(f2_add(a[0], a[1]), f2_add(c0, c1)), t0), t1)
    s2b = f2_mul(a[2], c1)
    s2a = f2_mul(a[2], c0)
    return (G.addxi(t0, s2b), mid, f2_add(t1, s2a))


def f6_mul_by_1(a, c1):
    return (f2_mul_xi(f2_mul(a[2], c1)), f2_mul(a[0], c1), f2_mul(a[1], c1))


def f6_inv(a):
    c0 = G.subxi(f2_sqr(a[0]), f2_mul(a[1], a[2]))
    c1 = f2_sub(f2_mul_xi(f2_sqr(a[2])), f2_mul(a[0], a[1]))
    c2 = f2_sub(f2_sqr(a[1]), f2_mul(a[0], a[2]))
    t = G.addxi(f2_mul(a[0], c0), f2_add(f2_mul(a[2], c1), f2_mul(a[1], c2)))
    ti = f2_inv(t)
    return (f2_mul(c0, ti), f2_mul(c1, ti), f2_mul(c2, ti))


# ---- Fp12 ------------------------------------------------------------------------------------------
def f6_zero():
    return (G.zero, G.zero, G.zero)


def f12_one():
    return ((G.one, G.zero, G.zero), f6_zero())


def f12_conj(a):
    return (a[0], f6_neg(a[1]))


def f6_add_mul_v(a, b):
    """a + v b"""
    return (G.addxi(a[0], b[2]), f2_add(a[1], b[0]), f2_add(a[2], b[1]))


def f6_sub_mul_v(a, b):
    return (G.subxi(a[0], b[2]), f2_sub(a[1], b[0]), f2_sub(a[2], b[1]))


def f12_mul(a, b):
    t0 = f6_mul(a[0], b[0])
    t1 = f6_mul(a[1], b[1])
    m = f6_mul(f6_add(a[0], a[1]), f6_add(b[0], b[1]))
    c1 = f6_sub(f6_sub(m, t0), t1)
    c0 = f6_add_mul_v(t0, t1)
    return (c0, c1)


def f12_sqr(a):
    ab = f6_mul(a[0], a[1])
    s = f6_mul(f6_add(a[0], a[1]), f6_add_mul_v(a[0], a[1]))
    c0 = f6_sub_mul_v(f6_sub(s, ab), ab)
    return (c0, f6_add(ab, ab))


def f12_mul_by_line(f, l0, l1, l2):
    aa = f6_mul_by_01(f[0], l0, l1)
    bb = f6_mul_by_1(f[1], l2)
    m = f6_mul_by_01(f6_add(f[0], f[1]), l0, f2_add(l1, l2))
    c1 = f6_sub(f6_sub(m, aa), bb)
    c0 = f6_add_mul_v(aa, bb)
    return (c0, c1)


def f12_inv(a):
    t0 = f6_sub_mul_v(f6_mul(a[0], a[0]), f6_mul(a[1], a[1]))
    ti = f6_inv(t0)
    return (f6_mul(a[0], ti), f6_neg(f6_mul(a[1], ti)))


FROB_GAMMA_INT = [i2_pow((1, 1), k * (P - 1) // 6) for k in range(6)]


def f12_frob(a):
    (a0, a2, a4), (a1, a3, a5) = a
    c = []
    for k, x in enumerate((a0, a1, a2, a3, a4, a5)):
        g = FROB_GAMMA_INT[k]
        xc = f2_conj(x)
        if g == (1, 0):
            c.append(xc)
        elif g[1] == 0:
            c.append(G.mulfp(xc, G.const(f"FROB{k}", g), 0))
        else:
            c.append(f2_mul(xc, G.const(f"FROB{k}", g)))
    return ((c[0], c[2], c[4]), (c[1], c[3], c[5]))


def f4_sqr(a, b):
    t0 = f2_sqr(a)
    t1 = f2_sqr(b)
    c0 = G.addxi(t0, t1)
    c1 = f2_sub(f2_sub(f2_sqr(f2_add(a, b)), t0), t1)
    return c0, c1


def f12_cyclotomic_sqr(f):
    (z0, z4, z3), (z2, z1, z5) = f
    t0, t1 = f4_sqr(z0, z1)
    z0 = f2_add(f2_dbl(f2_sub(t0, z0)), t0)
    z1 = f2_add(f2_dbl(f2_add(t1, z1)), t1)
    t0, t1 = f4_sqr(z2, z3)
    t2, t3 = f4_sqr(z4, z5)
    z4 = f2_add(f2_dbl(f2_sub(t0, z4)), t0)
    z5 = f2_add(f2_dbl(f2_add(t1, z5)), t1)
    t0 = f2_mul_xi(t3)
    z2 = f2_add(f2_dbl(f2_add(t0, z2)), t0)
    z3 = f2_add(f2_dbl(f2_sub(t2, z3)), t2)
    return ((z0, z4, z3), (z2, z1, z5))


def f12_cyc_pow_x(a):
    acc = a
    for b in range(62, -1, -1):
        acc = f12_cyclotomic_sqr(acc)
        if (X_ABS >> b) & 1:
            acc = f12_mul(acc, a)
    return f12_conj(acc)


# ---- Miller loop (csrc/bls_pairing.h) ----------------------------------------------------------------
def miller_dbl_step(f, T, pxy):
    X, Y, Z = T
    A = f2_sqr(X)
    B = f2_sqr(Y)
    C = f2_sqr(B)
    D = f2_dbl(f2_sub(f2_sub(f2_sqr(f2_add(X, B)), A), C))
    E = f2_mul3(A)
    Fq = f2_sqr(E)
    ZZ = f2_sqr(Z)
    Z3 = f2_dbl(f2_mul(Y, Z))
    l0 = f2_sub(f2_mul(E, X), f2_dbl(B))
    l1 = f2_neg(G.mulfp(f2_mul(E, ZZ), pxy, 0))
    l2 = G.mulfp(f2_mul(Z3, ZZ), pxy, 1)
    X3 = f2_sub(Fq, f2_dbl(D))
    C8 = f2_dbl(f2_dbl(f2_dbl(C)))
    Y3 = f2_sub(f2_mul(E, f2_sub(D, X3)), C8)
    return f12_mul_by_line(f, l0, l1, l2), (X3, Y3, Z3)


def miller_add_step(f, T, qx, qy, pxy):
    X, Y, Z = T
    Z1Z1 = f2_sqr(Z)
    U2 = f2_mul(qx, Z1Z1)
    S2 = f2_mul(f2_mul(qy, Z), Z1Z1)
    H = f2_sub(U2, X)
    HH = f2_sqr(H)
    I = f2_dbl(f2_dbl(HH))
    J = f2_mul(H, I)
    rr = f2_dbl(f2_sub(S2, Y))
    V = f2_mul(X, I)
    X3 = f2_sub(f2_sub(f2_sqr(rr), J), f2_dbl(V))
    Y3 = f2_sub(f2_mul(rr, f2_sub(V, X3)), f2_dbl(f2_mul(Y, J)))
    Z3 = f2_sub(f2_sub(f2_sqr(f2_add(Z, H)), Z1Z1), HH)
    l0 = f2_sub(f2_mul(rr, qx), f2_mul(qy, Z3))
    l1 = f2_neg(G.mulfp(rr, pxy, 0))
    l2 = G.mulfp(Z3, pxy, 1)
    return f12_mul_by_line(f, l0, l1, l2), (X3, Y3, Z3)


def miller_loop(pairs):
    """pairs: [(pxy, (qx, qy))]; f conjugated (x < 0)"""
    f = f12_one()
    Ts = [(q[0], q[1], G.one) for _, q in pairs]
    for b in range(62, -1, -1):
        if b != 62:
            f = f12_sqr(f)
        for k, (pxy, q) in enumerate(pairs):
            f, Ts[k] = miller_dbl_step(f, Ts[k], pxy)
        if (X_ABS >> b) & 1:
            for k, (pxy, q) in enumerate(pairs):
                f, Ts[k] = miller_add_step(f, Ts[k], q[0], q[1], pxy)
    return f12_conj(f)


def final_exponentiation(f):
    t = f12_mul(f12_conj(f), f12_inv(f))
    t = f12_mul(f12_frob(f12_frob(t)), t)
    a = f12_mul(f12_cyc_pow_x(t), f12_conj(t))
    a = f12_mul(f12_cyc_pow_x(a), f12_conj(a))
    b = f12_mul(f12_cyc_pow_x(a), f12_frob(a))
    c = f12_mul(f12_mul(f12_cyc_pow_x(f12_cyc_pow_x(b)), f12_frob(f12_frob(b))), f12_conj(b))
    return f12_mul(c, f12_mul(f12_cyclotomic_sqr(t), t))


G1_X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1

# inputs of part A: aggregate key as ONE register (x, y), H(m) and the signature as affine G2 coordinates
VERIFY_INPUTS = ["PXY", "HX", "HY", "SX", "SY"]
F12_NAMES = ["F%d" % k for k in range(6)]  # coefficient order c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2


def f12_flat(e):
    return [c for f6 in e for c in f6]


def f12_unflat(v):
    return ((v[0], v[1], v[2]), (v[3], v[4], v[5]))


def trace_part_a():
    """f = miller(P, H) * miller(-g1, S) and the Fp2 register whose c0 the final exponentiation inverts.
    Outputs: the 6 coefficients of f, then (d, 0)."""
    global G
    G = Graph()
    i = {n: G.inp(n) for n in VERIFY_INPUTS}
    g1n = G.const("G1_NEG", (G1_X, P - G1_Y))
    f = miller_loop([(i["PXY"], (i["HX"], i["HY"])), (g1n, (i["SX"], i["SY"]))])
    f12_inv(f)  # only to learn the inversion operand
    return G, f12_flat(f) + [G.inv_operand], [i[n] for n in VERIFY_INPUTS]


def trace_part_c():
    """final exponentiation of f given 1/d (in c0 of DINV); outputs the 6 coefficients of f^(3(p^12-1)/r)."""
    global G
    G = Graph()
    fin = [G.inp(n) for n in F12_NAMES]
    G.inv_result = G.inp("DINV")
    e = final_exponentiation(f12_unflat(fin))
    return G, f12_flat(e), fin + [G.inv_result]


# ------------------------------------------------------------------------------------------------
# schedule, allocate, encode
# ------------------------------------------------------------------------------------------------
class Program:
    pass


def needed_ops(g, outs):
    needed = [False] * len(g.nodes)
    stack = list(outs)
    while stack:
        v = stack.pop()
        if needed[v]:
            continue
        needed[v] = True
        kind, _, _, a, b, _ = g.nodes[v]
        if kind == K_OP:
            stack += [a, b]
    return [i for i in range(len(g.nodes)) if needed[i] and g.nodes[i][0] == K_OP]


# issue-cycle model of one round on one SIMD (v_mad_u64_u32 ~5 cycles per wave instruction,
# other VALU 2): Fp product ~2050, Fp addition ~210; plus LDS round trip + loop overhead
def round_cost(cls):
    c = 350
    if cls & (CLS_MUL | CLS_SQR | CLS_MULFP | CLS_NORM):
        c += 2050 * (3 if cls & CLS_MUL else 2)
        if cls & CLS_MUL:
            c += 5 * 210
        if cls & CLS_SQR:
            c += 3 * 210
        if cls & CLS_NORM:
            c += 210
    if cls & CLS_LIN:
        c += 2 * 260
    if cls & CLS_LINXI:
        c += 4 * 210
    return c


def make_program(g, ops, inputs, outputs, lanes, window):
    """list-schedule `ops` (node ids in trace order) into rounds of <= lanes operations; a round is either
    a product round or a linear round"""
    opset = set(ops)
    users = {i: [] for i in ops}
    ndeps = {}
    for i in ops:
        _, _, _, a, b, _ = g.nodes[i]
        d = 0
        for s in {a, b}:
            if s in opset:
                users[s].append(i)
                d += 1
        ndeps[i] = d
    is_prod = {i: g.nodes[i][1] in PROD_OPS for i in ops}
    cost = {i: (30 if is_prod[i] else 2) for i in ops}
    prio = {}
    for i in reversed(ops):
        prio[i] = cost[i] + max((prio[u] for u in users[i]), default=0)
    ready_prod = [i for i in ops if ndeps[i] == 0 and is_prod[i]]
    ready_lin = [i for i in ops if ndeps[i] == 0 and not is_prod[i]]
    rounds = []
    done = 0
    pos = {v: k for k, v in enumerate(ops)}
    scheduled = [False] * len(ops)
    head = 0
    while done < len(ops):
        while head < len(ops) and scheduled[head]:
            head += 1
        horizon = head + window
        el_lin = [i for i in ready_lin if pos[i] < horizon]
        el_prod = [i for i in ready_prod if pos[i] < horizon]
        if el_lin:
            el_lin.sort(key=lambda i: -prio[i])
            take = el_lin[:lanes]
            ts = set(take)
            ready_lin = [i for i in ready_lin if i not in ts]
        else:
            el_prod.sort(key=lambda i: -prio[i])
            take = el_prod[:lanes]
            ts = set(take)
            ready_prod = [i for i in ready_prod if i not in ts]
        assert take
        rounds.append(take)
        done += len(take)
        for i in take:
            scheduled[pos[i]] = True
            for u in users[i]:
                ndeps[u] -= 1
                if ndeps[u] == 0:
                    (ready_prod if is_prod[u] else ready_lin).append(u)
    # register allocation: constants and inputs pinned first, everything else linear scan
    const_nodes = sorted({s for i in ops for s in g.nodes[i][3:5] if g.nodes[s][0] == K_CONST} |
                         {o for o in outputs if g.nodes[o][0] == K_CONST})
    reg = {}
    nxt = 0
    for c in const_nodes:
        reg[c] = nxt
        nxt += 1
    for v in inputs:
        if v not in reg:
            reg[v] = nxt
            nxt += 1
    last_use = {}
    for r, take in enumerate(rounds):
        for i in take:
            for s in g.nodes[i][3:5]:
                last_use[s] = r
    pinned = set(const_nodes) | set(outputs)
    free = []
    release_at = {}
    for v in inputs:
        if v not in pinned and v in last_use:
            release_at.setdefault(last_use[v] + 1, []).append(reg[v])
    nreg = nxt
    enc_rounds = []
    for r, take in enumerate(rounds):
        for rr in release_at.pop(r, []):
            free.append(rr)
        row = []
        for i in take:
            if free:
                d = free.pop()
            else:
                d = nreg
                nreg += 1
            reg[i] = d
            if i not in pinned:
                release_at.setdefault(last_use.get(i, r) + 1, []).append(d)
            _, op, fl, a, b, _ = g.nodes[i]
            row.append((op, fl, d, reg[a], reg[b]))
        enc_rounds.append(row)
    pr = Program()
    pr.rounds = enc_rounds
    pr.nreg = nreg
    pr.const_regs = [(g.nodes[c][5], reg[c]) for c in const_nodes]
    pr.input_regs = [reg[v] for v in inputs]
    pr.output_regs = [reg[v] for v in outputs]
    pr.lanes = lanes
    pr.cls = []
    for row in enc_rounds:
        c = 0
        for op, *_ in row:
            c |= CLS_OF[op]
        pr.cls.append(c)
    pr.cycles = sum(round_cost(c) for c in pr.cls)
    pr.n_prod_rounds = sum(1 for c in pr.cls if c & 15)
    pr.n_lin_rounds = len(pr.cls) - pr.n_prod_rounds
    pr.n_prod = sum(len(r) for r, c in zip(enc_rounds, pr.cls) if c & 15)
    pr.n_lin = sum(len(r) for r, c in zip(enc_rounds, pr.cls) if not c & 15)
    pr.n_fp_products = sum({OP_MUL: 3, OP_SQR: 2, OP_MULFP: 2, OP_NORM: 2}.get(op, 0) for r in enc_rounds for op, *_ in r)
    return pr


def encode(pr):
    assert pr.nreg <= 256
    words = []
    for row in pr.rounds:
        for k in range(pr.lanes):
            if k < len(row):
                op, fl, d, a, b = row[k]
                words.append((op << 28) | (fl << 24) | (d << 16) | (a << 8) | b)
            else:
                words.append(0)
    return words


def simulate(pr, words, const_values, inputs):
    R = [(0, 0)] * pr.nreg
    for name, r in pr.const_regs:
        R[r] = const_values[name]
    for r, v in zip(pr.input_regs, inputs):
        R[r] = v
    L = pr.lanes
    for rd in range(len(words) // L):
        res = []
        for k in range(L):
            w = words[rd * L + k]
            op, fl, d, a, b = w >> 28, (w >> 24) & 15, (w >> 16) & 255, (w >> 8) & 255, w & 255
            if op != OP_NOP:
                res.append((d, i2_eval(op, fl, R[a], R[b])))
        for d, v in res:
            R[d] = v
    return [R[r] for r in pr.output_regs]


def eval_graph(g, targets, env):
    val = {}
    for i, (kind, op, fl, a, b, name) in enumerate(g.nodes):
        if kind == K_IN:
            val[i] = env[name]
        elif kind == K_CONST:
            val[i] = g.const_values[name]
        else:
            val[i] = i2_eval(op, fl, val[a], val[b])
    return [val[t] for t in targets]


def mont_limbs(v):
    m = v % P * (1 << 390) % P
    return [(m >> (30 * i)) & 0x3FFFFFFF for i in range(13)]


def emit(pa, pc, wa, wc, lanes, cva, cvc):
    def arr(name, vals, per=8, ty="unsigned int"):
        out = [f"static const {ty} {name}[{max(1, len(vals))}] = {{"]
        for i in range(0, len(vals), per):
            out.append("    " + ", ".join(("0x%08xu" % v) if ty == "unsigned int" else str(v) for v in vals[i:i + per]) + ",")
        out.append("};")
        return "\n".join(out)

    print("// GENERATED by tools/gen_bls_vm2.py -- do not edit.  Fp2 lane-group programs of the BLS pairing check.")
    print("// slot word: op[31:28] flags[27:24] dst[23:16] a[15:8] b[7:0]; one round = ECG_VM2_LANES words + one class byte.")
    print("#pragma once")
    print(f"#define ECG_VM2_LANES {lanes}")
    for tag, pr, w in (("A", pa, wa), ("C", pc, wc)):
        print(f"// part {tag}: {pr.n_prod} Fp2 products ({pr.n_fp_products} Fp products) in {pr.n_prod_rounds} rounds, "
              f"{pr.n_lin} linear ops in {pr.n_lin_rounds} rounds, {pr.nreg} registers, model {pr.cycles} cycles")
        print(f"#define ECG_VM2_{tag}_NREG {pr.nreg}")
        print(f"#define ECG_VM2_{tag}_ROUNDS {len(pr.rounds)}")
        print(f"#define ECG_VM2_{tag}_NIN {len(pr.input_regs)}")
        print(f"#define ECG_VM2_{tag}_NOUT {len(pr.output_regs)}")
        print(f"#define ECG_VM2_{tag}_FP_PRODUCTS {pr.n_fp_products}")
        print(arr(f"ECG_VM2_{tag}_IN", pr.input_regs))
        print(arr(f"ECG_VM2_{tag}_OUT", pr.output_regs))
        print(f"#define ECG_VM2_{tag}_NCONST {len(pr.const_regs)}")
        print(arr(f"ECG_VM2_{tag}_CONST_REG", [r for _, r in pr.const_regs]))
        cv = cva if tag == "A" else cvc
        print("// constant values: c0 then c1, 13 x 30-bit limbs each, Montgomery form (R = 2^390), one row per constant")
        print(arr(f"ECG_VM2_{tag}_CONST_VAL", [x for n, _ in pr.const_regs for c in cv[n] for x in mont_limbs(c)], per=13))
        print(arr(f"ECG_VM2_{tag}_CLS", pr.cls, per=32, ty="unsigned char"))
        print(arr(f"ECG_VM2_{tag}_PROG", w))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=16)
    ap.add_argument("--window", type=int, default=200)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--stats", action="store_true")
    args = ap.parse_args()
    sys.setrecursionlimit(10000)
    ga, outs_a, ins_a = trace_part_a()
    pa = make_program(ga, needed_ops(ga, outs_a), ins_a, outs_a, args.lanes, args.window)
    gc, outs_c, ins_c = trace_part_c()
    pc = make_program(gc, needed_ops(gc, outs_c), ins_c, outs_c, args.lanes, args.window)
    wa, wc = encode(pa), encode(pc)
    if args.stats or args.check:
        for tag, pr in (("A", pa), ("C", pc)):
            sys.stderr.write(f"part {tag}: prod {pr.n_prod} in {pr.n_prod_rounds} rounds ({pr.n_prod / max(1, pr.n_prod_rounds):.1f}/round), "
                             f"lin {pr.n_lin} in {pr.n_lin_rounds} rounds ({pr.n_lin / max(1, pr.n_lin_rounds):.1f}/round), "
                             f"fp products {pr.n_fp_products}, nreg {pr.nreg}, model cycles {pr.cycles}\n")
        tot = pa.cycles + pc.cycles
        sys.stderr.write(f"model: {tot} cycles per wave of {64 // args.lanes} tuples -> "
                         f"{tot * (65536 / (64 // args.lanes)) / 1024 / 2.4e9 * 1e3:.1f} ms per 65536 tuples (1024 SIMDs, 2.4 GHz)\n")
    if args.check:
        rnd = random.Random(1)
        for _ in range(2):
            env = {nm: (rnd.randrange(P), rnd.randrange(P)) for nm in VERIFY_INPUTS}
            mid = simulate(pa, wa, ga.const_values, [env[nm] for nm in VERIFY_INPUTS])
            assert mid == eval_graph(ga, outs_a, env)
            dinv = (pow(mid[6][0], P - 2, P), 0)
            env_c = dict(zip(F12_NAMES, mid[:6]))
            env_c["DINV"] = dinv
            got = simulate(pc, wc, gc.const_values, mid[:6] + [dinv])
            assert got == eval_graph(gc, outs_c, env_c), "encoded program disagrees with the traced expression"
        sys.stderr.write("check ok\n")
        return
    if not args.stats:
        emit(pa, pc, wa, wc, args.lanes, ga.const_values, gc.const_values)


if __name__ == "__main__":
    main()
