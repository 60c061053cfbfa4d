#!/usr/bin/env python3
"""Generates ethereum_consensus_amd/csrc/bls_vm_prog.h: the lane-group programs of the BLS pairing
check for the gfx950 "field VM" kernels (csrc/bls_vm.h, bls_vm.hip).

Why: an Fp12 is 12 x 13 dwords.  One lane per pairing keeps every Fp12 temporary in the private
segment and the kernel ends up bound by scratch traffic to HBM (profiles/r01b_bls_occupancy_probe.txt).
Here a GROUP of G lanes shares one pairing: the computation is traced once into a straight-line
program over Fp values (mul / add / sub), list-scheduled into rounds of at most G independent
operations of one kind, and register-allocated onto an LDS-resident register file (13 dwords per
register per tuple).  Every lane of a wave executes the same instruction stream (one Fp product, or
one Fp addition, per round) on its own operands, so there is no divergence, no scratch, and the
working set of a tuple (a few KB) lives in LDS.

The program is the same algorithm as csrc/bls_pairing.h (2-pair Miller loop over |x| with shared
squaring on the M-twist, final exponentiation with Granger-Scott squarings), traced symbolically.
The single Fp inversion of the final exponentiation is a long sequential chain with no
parallelism, so the trace is cut there: part A (Miller loop .. norm), a lane-per-tuple inversion
kernel, part C (rest of the final exponentiation, == 1 test).

Self-contained (no import of oracle/); `--check` simulates the ENCODED programs on random inputs
with Python integers and compares with a direct evaluation of the same trace.

    python tools/gen_bls_vm.py [--lanes 16] > ethereum_consensus_amd/csrc/bls_vm_prog.h
"""
import argparse
import random
import sys

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
X_ABS = 0xD201000000010000

# ------------------------------------------------------------------------------------------------
# expression graph over Fp
# ------------------------------------------------------------------------------------------------
OP_IN, OP_CONST, OP_MUL, OP_ADD, OP_SUB, OP_INV = "in", "const", "mul", "add", "sub", "inv"


class Graph:
    def __init__(self):
        self.nodes = []  # (op, a, b, name)
        self.memo = {}
        self.inv_operand = None
        self.inv_result = None
        self.zero = self.const("ZERO", 0)
        self.one = self.const("ONE", 1)

    def _new(self, op, a=None, b=None, name=None):
        key = (op, a, b, name)
        if key in self.memo:
            return self.memo[key]
        self.nodes.append(key)
        self.memo[key] = len(self.nodes) - 1
        return len(self.nodes) - 1

    def inp(self, name):
        return self._new(OP_IN, name=name)

    def const(self, name, value):
        i = self._new(OP_CONST, name=name)
        if not hasattr(self, "const_values"):
            self.const_values = {}
        self.const_values[name] = value % P
        return i

    def mul(self, a, b):
        if a == self.zero or b == self.zero:
            return self.zero
        if a == self.one:
            return b
        if b == self.one:
            return a
        if a > b:
            a, b = b, a
        return self._new(OP_MUL, a, b)

    def add(self, a, b):
        if a == self.zero:
            return b
        if b == self.zero:
            return a
        if a > b:
            a, b = b, a
        return self._new(OP_ADD, a, b)

    def sub(self, a, b):
        if b == self.zero:
            return a
        if a == b:
            return self.zero
        return self._new(OP_SUB, a, b)

    def neg(self, a):
        return self.sub(self.zero, a)

    def inv(self, a):
        """The one inversion of the pairing check.  Part A records its operand and stops; part C receives
        the inverse as an input."""
        self.inv_operand = a
        if self.inv_result is not None:
            return self.inv_result
        return self._new(OP_INV, a)


G = None  # the graph being traced (module-level so that the tower code below reads naturally)


# ---- Fp2 -------------------------------------------------------------------------------------------
def f2(c0, c1):
    return (c0, c1)


def f2_zero():
    return (G.zero, G.zero)


def f2_one():
    return (G.one, G.zero)


def f2_add(a, b):
    return (G.add(a[0], b[0]), G.add(a[1], b[1]))


def f2_sub(a, b):
    return (G.sub(a[0], b[0]), G.sub(a[1], b[1]))


def f2_neg(a):
    return (G.neg(a[0]), G.neg(a[1]))


def f2_dbl(a):
    return f2_add(a, a)


def f2_conj(a):
    return (a[0], G.neg(a[1]))


def f2_mul_xi(a):
    return (G.sub(a[0], a[1]), G.add(a[0], a[1]))


def f2_mul_fp(a, k):
    return (G.mul(a[0], k), G.mul(a[1], k))


def f2_mul(a, b):
    if a[1] == G.zero and b[1] == G.zero:
        return (G.mul(a[0], b[0]), G.zero)
    if b[1] == G.zero:
        return f2_mul_fp(a, b[0])
    if a[1] == G.zero:
        return f2_mul_fp(b, a[0])
    t0 = G.mul(a[0], b[0])
    t1 = G.mul(a[1], b[1])
    t2 = G.mul(G.add(a[0], a[1]), G.add(b[0], b[1]))
    return (G.sub(t0, t1), G.sub(G.sub(t2, t0), t1))


def f2_sqr(a):
    t0 = G.mul(G.add(a[0], a[1]), G.sub(a[0], a[1]))
    t1 = G.mul(a[0], a[1])
    return (t0, G.add(t1, t1))


def f2_mul3(a):
    return f2_add(f2_dbl(a), a)


def f2_inv(a):
    d = G.inv(G.add(G.mul(a[0], a[0]), G.mul(a[1], a[1])))
    return (G.mul(a[0], d), G.neg(G.mul(a[1], d)))


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
    c0 = f2_add(t0, f2_mul_xi(f2_sub(f2_sub(m12, t1), t2)))
    c1 = f2_add(f2_sub(f2_sub(m01, t0), t1), f2_mul_xi(t2))
    c2 = f2_add(f2_sub(f2_sub(m02, t0), t2), t1)
    return (c0, c1, c2)


def f6_mul_by_01(a, c0, c1):
    t0 = f2_mul(a[0], c0)
    t1 = f2_mul(a[1], c1)
    mid = f2_sub(f2_sub(f2_mul(f2_add(a[0], a[1]), f2_add(c0, c1)), t0), t1)
    s2b = f2_mul(a[2], c1)
    s2a = f2_mul(a[2], c0)
    return (f2_add(t0, f2_mul_xi(s2b)), mid, f2_add(t1, s2a))


def f6_mul_by_1(a, c1):
    return (f2_mul_xi(f2_mul(a[2], c1)), f2_mul(a[0], c1), f2_mul(a[1], c1))


def f6_inv(a):
    c0 = f2_sub(f2_sqr(a[0]), f2_mul_xi(f2_mul(a[1], a[2])))
    c1 = f2_sub(f2_mul_xi(f2_sqr(a[2])), f2_mul(a[0], a[1]))
    c2 = f2_sub(f2_sqr(a[1]), f2_mul(a[0], a[2]))
    t = f2_add(f2_mul(a[0], c0), f2_mul_xi(f2_add(f2_mul(a[2], c1), f2_mul(a[1], c2))))
    ti = f2_inv(t)
    return (f2_mul(c0, ti), f2_mul(c1, ti), f2_mul(c2, ti))


# ---- Fp12 ------------------------------------------------------------------------------------------
def f12_one():
    return ((f2_one(), f2_zero(), f2_zero()), (f2_zero(), f2_zero(), f2_zero()))


def f12_conj(a):
    return (a[0], f6_neg(a[1]))


def f12_mul(a, b):
    t0 = f6_mul(a[0], b[0])
    t1 = f6_mul(a[1], b[1])
    m = f6_mul(f6_add(a[0], a[1]), f6_add(b[0], b[1]))
    c1 = f6_sub(f6_sub(m, t0), t1)
    c0 = f6_add(t0, f6_mul_v(t1))
    return (c0, c1)


def f12_sqr(a):
    ab = f6_mul(a[0], a[1])
    s = f6_mul(f6_add(a[0], a[1]), f6_add(a[0], f6_mul_v(a[1])))
    c0 = f6_sub(f6_sub(s, ab), f6_mul_v(ab))
    return (c0, f6_add(ab, ab))


def f12_mul_by_line(f, l0, l1, l2):
    aa = f6_mul_by_01(f[0], l0, l1)
    bb = f6_mul_by_1(f[1], l2)
    m = f6_mul_by_01(f6_add(f[0], f[1]), l0, f2_add(l1, l2))
    c1 = f6_sub(f6_sub(m, aa), bb)
    c0 = f6_add(aa, f6_mul_v(bb))
    return (c0, c1)


def f12_inv(a):
    t0 = f6_sub(f6_mul(a[0], a[0]), f6_mul_v(f6_mul(a[1], a[1])))
    ti = f6_inv(t0)
    return (f6_mul(a[0], ti), f6_neg(f6_mul(a[1], ti)))


def f2_pow_int(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = ((r[0] * a[0] - r[1] * a[1]) % P, (r[0] * a[1] + r[1] * a[0]) % P)
        a = ((a[0] * a[0] - a[1] * a[1]) % P, (2 * a[0] * a[1]) % P)
        e >>= 1
    return r


FROB_GAMMA_INT = [f2_pow_int((1, 1), k * (P - 1) // 6) for k in range(6)]


def frob_consts():
    out = []
    for k in range(6):
        g = FROB_GAMMA_INT[k]
        out.append((G.const(f"FROB{k}_C0", g[0]) if g[0] not in (0, 1) else (G.zero if g[0] == 0 else G.one),
                    G.const(f"FROB{k}_C1", g[1]) if g[1] not in (0, 1) else (G.zero if g[1] == 0 else G.one)))
    return out


def f12_frob(a):
    gam = frob_consts()
    (a0, a2, a4), (a1, a3, a5) = a
    c = [f2_mul(f2_conj(x), gam[k]) for k, x in enumerate((a0, a1, a2, a3, a4, a5))]
    return ((c[0], c[2], c[4]), (c[1], c[3], c[5]))


def f4_sqr(a, b):
    t0 = f2_sqr(a)
    t1 = f2_sqr(b)
    c0 = f2_add(f2_mul_xi(t1), t0)
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


# ---- Miller loop -----------------------------------------------------------------------------------
def miller_dbl_step(f, T, px, py):
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
    l1 = f2_neg(f2_mul_fp(f2_mul(E, ZZ), px))
    l2 = f2_mul_fp(f2_mul(Z3, ZZ), py)
    X3 = f2_sub(Fq, f2_dbl(D))
    C8 = f2_dbl(f2_dbl(f2_dbl(C)))
    Y3 = f2_sub(f2_mul(E, f2_sub(D, X3)), C8)
    return f12_mul_by_line(f, l0, l1, l2), (X3, Y3, Z3)


def miller_add_step(f, T, qx, qy, px, py):
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
    l1 = f2_neg(f2_mul_fp(rr, px))
    l2 = f2_mul_fp(Z3, py)
    return f12_mul_by_line(f, l0, l1, l2), (X3, Y3, Z3)


def miller_loop(pairs):
    """pairs: [(px, py, (qx, qy))]; f conjugated (x < 0)"""
    f = f12_one()
    Ts = [(q[0], q[1], f2_one()) for _, _, q in pairs]
    for b in range(62, -1, -1):
        if b != 62:
            f = f12_sqr(f)
        for k, (px, py, q) in enumerate(pairs):
            f, Ts[k] = miller_dbl_step(f, Ts[k], px, py)
        if (X_ABS >> b) & 1:
            for k, (px, py, q) in enumerate(pairs):
                f, Ts[k] = miller_add_step(f, Ts[k], q[0], q[1], px, py)
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

# order of the inputs of the verify program: aggregate key (affine G1), H(m) (affine G2), signature (affine G2)
VERIFY_INPUTS = ["PX", "PY", "HX0", "HX1", "HY0", "HY1", "SX0", "SX1", "SY0", "SY1"]


F12_NAMES = ["F%d" % k for k in range(12)]  # coefficient order c0.c0.c0, c0.c0.c1, c0.c1.c0 .. c1.c2.c1


def f12_flat(e):
    return [c for f6 in e for f2_ in f6 for c in f2_]


def f12_unflat(v):
    return (((v[0], v[1]), (v[2], v[3]), (v[4], v[5])), ((v[6], v[7]), (v[8], v[9]), (v[10], v[11])))


def trace_part_a():
    """f = miller(P, H) * miller(-g1, S) and the Fp element d whose inverse the final exponentiation
    needs (the norm down to Fp of f).  Outputs: 12 coefficients of f, then d."""
    global G
    G = Graph()
    i = {n: G.inp(n) for n in VERIFY_INPUTS}
    gx = G.const("G1_X", G1_X)
    gny = G.const("G1_NEG_Y", P - G1_Y)
    pairs = [(i["PX"], i["PY"], ((i["HX0"], i["HX1"]), (i["HY0"], i["HY1"]))),
             (gx, gny, ((i["SX0"], i["SX1"]), (i["SY0"], i["SY1"])))]
    f = miller_loop(pairs)
    f12_inv(f)  # only to learn the inversion operand
    return G, f12_flat(f) + [G.inv_operand], [i[n] for n in VERIFY_INPUTS]


def trace_part_c():
    """final exponentiation of f given 1/d; outputs the 12 coefficients of f^(3(p^12-1)/r)."""
    global G
    G = Graph()
    fin = [G.inp(n) for n in F12_NAMES]
    G.inv_result = G.inp("DINV")
    e = final_exponentiation(f12_unflat(fin))
    return G, f12_flat(e), fin + [G.inv_result]


# ------------------------------------------------------------------------------------------------
# cut at the inversion, schedule, allocate, encode
# ------------------------------------------------------------------------------------------------
class Program:
    pass


def needed_ops(g, outs):
    n = len(g.nodes)
    needed = [False] * n
    stack = list(outs)
    while stack:
        v = stack.pop()
        if needed[v]:
            continue
        needed[v] = True
        op, a, b, _ = g.nodes[v]
        if op in (OP_MUL, OP_ADD, OP_SUB):
            stack += [a, b]
    return [i for i in range(n) if needed[i] and g.nodes[i][0] in (OP_MUL, OP_ADD, OP_SUB)]


def make_program(g, ops, inputs, outputs, lanes, window):
    """list-schedule `ops` (node ids, topologically ordered) into homogeneous rounds of <= lanes ops"""
    opset = set(ops)
    users = {i: [] for i in ops}
    ndeps = {}
    for i in ops:
        op, a, b, _ = g.nodes[i]
        d = 0
        for s in {a, b}:
            if s in opset:
                users[s].append(i)
                d += 1
        ndeps[i] = d
    # priority = longest path to a sink, in cost units (a product ~ 6 additions)
    cost = {OP_MUL: 6, OP_ADD: 1, OP_SUB: 1}
    prio = {}
    for i in reversed(ops):
        prio[i] = cost[g.nodes[i][0]] + max((prio[u] for u in users[i]), default=0)
    ready_mul = [i for i in ops if ndeps[i] == 0 and g.nodes[i][0] == OP_MUL]
    ready_lin = [i for i in ops if ndeps[i] == 0 and g.nodes[i][0] != OP_MUL]
    rounds = []
    done = 0
    pos = {v: k for k, v in enumerate(ops)}  # position in trace order (a low-register-pressure order)
    scheduled = [False] * len(ops)
    head = 0  # first unscheduled position
    while done < len(ops):
        while head < len(ops) and scheduled[head]:
            head += 1
        horizon = head + window  # only operations this close to the oldest pending one may issue:
        # bounds how far the schedule runs ahead of the trace order, hence the live ranges / LDS registers
        el_lin = [i for i in ready_lin if pos[i] < horizon]
        el_mul = [i for i in ready_mul if pos[i] < horizon]
        # cheap additions first (they unlock products); a product round when no addition is ready
        if el_lin:
            el_lin.sort(key=lambda i: -prio[i])
            take = el_lin[:lanes]
            kind = "lin"
            ts = set(take)
            ready_lin = [i for i in ready_lin if i not in ts]
        else:
            el_mul.sort(key=lambda i: -prio[i])
            take = el_mul[:lanes]
            kind = "mul"
            ts = set(take)
            ready_mul = [i for i in ready_mul if i not in ts]
        assert take
        rounds.append((kind, take))
        done += len(take)
        for i in take:
            scheduled[pos[i]] = True
            for u in users[i]:
                ndeps[u] -= 1
                if ndeps[u] == 0:
                    (ready_mul if g.nodes[u][0] == OP_MUL else ready_lin).append(u)
    # register allocation: constants and inputs pinned first, everything else linear scan
    const_nodes = sorted({s for i in ops for s in g.nodes[i][1:3] if g.nodes[s][0] == OP_CONST} |
                         {o for o in outputs if g.nodes[o][0] == OP_CONST})
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
    for r, (_, take) in enumerate(rounds):
        for i in take:
            for s in g.nodes[i][1:3]:
                last_use[s] = r
    pinned = set(const_nodes) | set(outputs)
    free = []
    release_at = {}
    for v in inputs:
        if v not in pinned and v in last_use:
            release_at.setdefault(last_use[v] + 1, []).append(reg[v])
        elif v not in pinned and v not in last_use:
            pass  # unused input keeps its register
    nreg = nxt
    enc_rounds = []
    for r, (kind, take) in enumerate(rounds):
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
                lu = last_use.get(i)
                if lu is None:
                    release_at.setdefault(r + 1, []).append(d)  # dead value (should not happen)
                else:
                    release_at.setdefault(lu + 1, []).append(d)
            op, a, b, _ = g.nodes[i]
            row.append((op, d, reg[a], reg[b]))
        enc_rounds.append((kind, row))
    pr = Program()
    pr.rounds = enc_rounds
    pr.nreg = nreg
    pr.const_regs = [(g.nodes[c][3], reg[c]) for c in const_nodes]
    pr.input_regs = [reg[v] for v in inputs]
    pr.output_regs = [reg[v] for v in outputs]
    pr.lanes = lanes
    pr.n_mul = sum(len(t) for k, t in rounds if k == "mul")
    pr.n_lin = sum(len(t) for k, t in rounds if k == "lin")
    pr.n_mul_rounds = sum(1 for k, _ in rounds if k == "mul")
    pr.n_lin_rounds = sum(1 for k, _ in rounds if k == "lin")
    return pr


OPC = {"nop": 0, OP_MUL: 1, OP_ADD: 2, OP_SUB: 3}


def encode(pr):
    """one u32 per lane slot: op[31:30] dst[29:20] a[19:10] b[9:0]"""
    assert pr.nreg <= 1024
    words = []
    for kind, row in pr.rounds:
        for k in range(pr.lanes):
            if k < len(row):
                op, d, a, b = row[k]
                words.append((OPC[op] << 30) | (d << 20) | (a << 10) | b)
            else:
                words.append(0)
    return words


def simulate(pr, words, const_values, inputs):
    R = [0] * pr.nreg
    for name, r in pr.const_regs:
        R[r] = const_values[name]
    for r, v in zip(pr.input_regs, inputs):
        R[r] = v
    L = pr.lanes
    for rd in range(len(words) // L):
        res = []
        for k in range(L):
            w = words[rd * L + k]
            op, d, a, b = w >> 30, (w >> 20) & 1023, (w >> 10) & 1023, w & 1023
            if op == 1:
                res.append((d, R[a] * R[b] % P))
            elif op == 2:
                res.append((d, (R[a] + R[b]) % P))
            elif op == 3:
                res.append((d, (R[a] - R[b]) % P))
        for d, v in res:
            R[d] = v
    return [R[r] for r in pr.output_regs]


def eval_graph(g, targets, env):
    val = {}
    for i, (op, a, b, name) in enumerate(g.nodes):
        if op == OP_IN:
            val[i] = env[name]
        elif op == OP_CONST:
            val[i] = g.const_values[name]
        elif op == OP_MUL:
            val[i] = val[a] * val[b] % P
        elif op == OP_ADD:
            val[i] = (val[a] + val[b]) % P
        elif op == OP_SUB:
            val[i] = (val[a] - val[b]) % P
        elif op == OP_INV:
            val[i] = pow(val[a], P - 2, P)
    return [val[t] for t in targets]


def mont_limbs(v):
    m = v % P * (1 << 390) % P
    return [(m >> (30 * i)) & 0x3FFFFFFF for i in range(13)]


def emit(pa, pc, wa, wc, lanes, cva, cvc):
    def arr(name, vals, per=8):
        out = [f"static const unsigned int {name}[{len(vals)}] = {{"]
        for i in range(0, len(vals), per):
            out.append("    " + ", ".join("0x%08xu" % v for v in vals[i:i + per]) + ",")
        out.append("};")
        return "\n".join(out)

    print("// GENERATED by tools/gen_bls_vm.py -- do not edit.  Lane-group programs of the BLS pairing check.")
    print("// slot word: op[31:30] (0 nop, 1 mul, 2 add, 3 sub) dst[29:20] a[19:10] b[9:0]; one round = VM_LANES words.")
    print("#pragma once")
    print(f"#define ECG_VM_LANES {lanes}")
    for tag, pr, w in (("A", pa, wa), ("C", pc, wc)):
        print(f"// part {tag}: {pr.n_mul} products in {pr.n_mul_rounds} rounds, {pr.n_lin} additions in {pr.n_lin_rounds} rounds, "
              f"{pr.nreg} registers")
        print(f"#define ECG_VM_{tag}_NREG {pr.nreg}")
        print(f"#define ECG_VM_{tag}_ROUNDS {len(pr.rounds)}")
        print(f"#define ECG_VM_{tag}_NIN {len(pr.input_regs)}")
        print(f"#define ECG_VM_{tag}_NOUT {len(pr.output_regs)}")
        print(arr(f"ECG_VM_{tag}_IN", pr.input_regs))
        print(arr(f"ECG_VM_{tag}_OUT", pr.output_regs))
        print(f"#define ECG_VM_{tag}_NCONST {len(pr.const_regs)}")
        print(f"static const char* const ECG_VM_{tag}_CONST_NAME[{len(pr.const_regs)}] = {{" +
              ", ".join('"%s"' % n for n, _ in pr.const_regs) + "};")
        print(arr(f"ECG_VM_{tag}_CONST_REG", [r for _, r in pr.const_regs]))
        cv = cva if tag == "A" else cvc
        print(f"// constant values: 13 x 30-bit limbs, Montgomery form (R = 2^390), one row per constant")
        print(arr(f"ECG_VM_{tag}_CONST_VAL", [w for n, _ in pr.const_regs for w in mont_limbs(cv[n])], per=13))
        print(arr(f"ECG_VM_{tag}_PROG", w))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=16)
    ap.add_argument("--window", type=int, default=400)
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
            sys.stderr.write(f"part {tag}: mul {pr.n_mul} in {pr.n_mul_rounds} rounds ({pr.n_mul / max(1, pr.n_mul_rounds):.1f}/round), "
                             f"lin {pr.n_lin} in {pr.n_lin_rounds} rounds ({pr.n_lin / max(1, pr.n_lin_rounds):.1f}/round), "
                             f"nreg {pr.nreg}, in {len(pr.input_regs)}, out {len(pr.output_regs)}\n")
    if args.check:
        rnd = random.Random(1)
        for _ in range(2):
            env = {nm: rnd.randrange(P) for nm in VERIFY_INPUTS}
            mid = simulate(pa, wa, ga.const_values, [env[nm] for nm in VERIFY_INPUTS])
            assert mid == eval_graph(ga, outs_a, env)
            dinv = pow(mid[12], P - 2, P)
            env_c = dict(zip(F12_NAMES, mid[:12]))
            env_c["DINV"] = dinv
            got = simulate(pc, wc, gc.const_values, mid[:12] + [dinv])
            assert got == eval_graph(gc, outs_c, env_c), "encoded program disagrees with the traced expression"
        sys.stderr.write("check ok\n")
        return
    if not args.stats:
        emit(pa, pc, wa, wc, args.lanes, ga.const_values, gc.const_values)


if __name__ == "__main__":
    main()
