// Fp6 / Fp12 tower arithmetic for the BLS12-381 pairing on gfx950:
//   Fp6 = Fp2[v]/(v^3 - xi), xi = 1 + i;   Fp12 = Fp6[w]/(w^2 - v).
// (blst's fp12_tower.c layer under /root/reference/ethereum-consensus/src/crypto/bls.rs:69-71,
// 102-106, 122-126 -- the verify calls.)  Products are sums of half-products with one Montgomery reduction per
// coefficient over lazy operands (bls_fp.h fp_sumprod): schoolbook in Fp2 and Fp6, Karatsuba only at the Fp12 level.
// An Fp12 is 12 x 13 dwords; the accumulators of the Miller loop and of the final exponentiation are kept as local
// values so that they stay in the 512-entry register file across an iteration.
#pragma once
#include "bls_fp.h"

// Call structure of the lane kernels (every step measured on 65 536 tuples).  The Fp6-level routines and
// the sums of products under them are inline: their operands are SSA values in VGPRs / AGPRs, and an out-of-line routine
// that keeps values live across calls must save callee-saved registers on every entry (229-282 dwords per Fp6 product
// when those were calls: the bulk of the round-1 private-segment traffic).  The Fp12-level routines are out of line -- one
// copy of each per kernel, operands copied through private-segment pointers (ecg_priv_load) -- except where a loop runs
// them on a register-resident accumulator: the doubling iteration of the Miller loop (accumulator squaring, doubling step,
// line product) and the cyclotomic squaring inside the exponentiations by x.  Everything inline was slower (one 2.8 MB
// function, the register allocator spills more than the calls cost).
#define ECG_FP6_FN ECG_HD
#define ECG_FP12_FN ECG_HD_NOINLINE
#define ECG_MILLER_DBL_FN ECG_HD

namespace ecg {

// Fp2 products are inlined into the Fp6-level routines (only the Fp products underneath are calls), so the
// Karatsuba temporaries of a routine are SSA values in VGPRs/AGPRs instead of private-segment objects: measured
// -40 % HBM traffic and 71 -> 46 ms on the pairing kernel (profiles/r01k_*), which is bound by that traffic.
ECG_HD Fp2 fp2_mulx(const Fp2& a, const Fp2& b) { return fp2_mul(a, b); }
ECG_HD Fp2 fp2_sqrx(const Fp2& a) { return fp2_sqr(a); }

// ---------------------------------------------------------------------------------------------
// Fp6
// ---------------------------------------------------------------------------------------------
ECG_HD Fp6 fp6_zero() { return Fp6{fp2_zero(), fp2_zero(), fp2_zero()}; }
ECG_HD Fp6 fp6_one() { return Fp6{fp2_one(), fp2_zero(), fp2_zero()}; }
ECG_HD void fp6_add(Fp6& r, const Fp6& a, const Fp6& b) {
    r.c0 = fp2_add(a.c0, b.c0);
    r.c1 = fp2_add(a.c1, b.c1);
    r.c2 = fp2_add(a.c2, b.c2);
}
ECG_HD void fp6_sub(Fp6& r, const Fp6& a, const Fp6& b) {
    r.c0 = fp2_sub(a.c0, b.c0);
    r.c1 = fp2_sub(a.c1, b.c1);
    r.c2 = fp2_sub(a.c2, b.c2);
}
ECG_HD void fp6_neg(Fp6& r, const Fp6& a) {
    r.c0 = fp2_neg(a.c0);
    r.c1 = fp2_neg(a.c1);
    r.c2 = fp2_neg(a.c2);
}
// multiply by v: (a0 + a1 v + a2 v^2) v = xi a2 + a0 v + a1 v^2
ECG_HD void fp6_mul_v(Fp6& r, const Fp6& a) {
    Fp2 t = fp2_mul_xi(a.c2);
    r.c2 = a.c1;
    r.c1 = a.c0;
    r.c0 = t;
}
// Schoolbook with lazy reduction: every Fp2 coefficient of the result is ONE sum of three Fp2 products (xi folded
// into the a operands, the minus signs of the Fp2 products into lazily negated b components): 36 half-products and 6
// reductions, no linear operation on any result (measured 48.7 k cycles against 65.0 k for Karatsuba over 18 reduced
// products and its 48 modular additions, profiles/r01zf_fpbench.txt).  Components of a < KA p, of b < KB p; a sum is at
// most 6 * (2 KA)(KB) p^2.
template <int KA, int KB>
ECG_HD void fp6_mul_lazy(Fp6& r, const Fp2& a0, const Fp2& a1, const Fp2& a2, const Fp2& b0, const Fp2& b1, const Fp2& b2) {
    static_assert(12 * KA * KB < 632, "sum of products would not reduce below 2p");
    const Fp2 xa1 = fp2_mul_xi_lazy<KA>(a1), xa2 = fp2_mul_xi_lazy<KA>(a2);
    const Fp n0 = fp_neg_lazy<KB>(b0.c1), n1 = fp_neg_lazy<KB>(b1.c1), n2 = fp_neg_lazy<KB>(b2.c1);
    Fp2 c0, c1, c2;
    {
        const Fp2 x[3] = {a0, xa1, xa2};
        const Fp2 y[3] = {b0, b2, b1};
        const Fp ny[3] = {n0, n2, n1};
        c0 = fp2_sumprod<3>(x, y, ny);
    }
    {
        const Fp2 x[3] = {a0, a1, xa2};
        const Fp2 y[3] = {b1, b0, b2};
        const Fp ny[3] = {n1, n0, n2};
        c1 = fp2_sumprod<3>(x, y, ny);
    }
    {
        const Fp2 x[3] = {a0, a1, a2};
        const Fp2 y[3] = {b2, b1, b0};
        const Fp ny[3] = {n2, n1, n0};
        c2 = fp2_sumprod<3>(x, y, ny);
    }
    r.c0 = c0;
    r.c1 = c1;
    r.c2 = c2;
}
// r may alias a or b (operands are loaded before the result is stored).
ECG_FP6_FN void fp6_mul(Fp6& r, const Fp6& a, const Fp6& b) {
    const Fp2 a0 = a.c0, a1 = a.c1, a2 = a.c2, b0 = b.c0, b1 = b.c1, b2 = b.c2;
    fp6_mul_lazy<2, 2>(r, a0, a1, a2, b0, b1, b2);
}
// r = (a0 + a1)(b0 + b1): the middle product of the Fp12 Karatsuba, sums formed on the fly (< 4p).  r must not alias.
ECG_FP6_FN void fp6_mul_sums(Fp6& r, const Fp6& a0, const Fp6& a1, const Fp6& b0, const Fp6& b1) {
    fp6_mul_lazy<4, 4>(r, fp2_add_lazy(a0.c0, a1.c0), fp2_add_lazy(a0.c1, a1.c1), fp2_add_lazy(a0.c2, a1.c2), fp2_add_lazy(b0.c0, b1.c0),
                       fp2_add_lazy(b0.c1, b1.c1), fp2_add_lazy(b0.c2, b1.c2));
}
// r = (a0 + a1)(a0 + v a1): the first product of the complex squaring.  First operand < 4p; second: a0.c0 + xi a1.c2
// < 2p + 4p.  r must not alias.
ECG_FP6_FN void fp6_mul_sqr_sums(Fp6& r, const Fp6& a0, const Fp6& a1) {
    fp6_mul_lazy<4, 6>(r, fp2_add_lazy(a0.c0, a1.c0), fp2_add_lazy(a0.c1, a1.c1), fp2_add_lazy(a0.c2, a1.c2),
                       fp2_add_lazy(a0.c0, fp2_mul_xi_lazy<2>(a1.c2)), fp2_add_lazy(a0.c1, a1.c0), fp2_add_lazy(a0.c2, a1.c1));
}
// Karatsuba recombination in one pass: r1 = m - t0 - t1, r0 = t0 + v t1.  r0 / r1 may alias m, t0, t1
// component-wise (every component is read before it is written).
ECG_FP6_FN void fp12_karatsuba_combine(Fp6& r0, Fp6& r1, const Fp6& m, const Fp6& t0, const Fp6& t1) {
    const Fp2 x0 = t0.c0, x1 = t0.c1, x2 = t0.c2, y0 = t1.c0, y1 = t1.c1, y2 = t1.c2;
    const Fp2 m0 = m.c0, m1 = m.c1, m2 = m.c2;
    r1.c0 = fp2_sub(fp2_sub(m0, x0), y0);
    r1.c1 = fp2_sub(fp2_sub(m1, x1), y1);
    r1.c2 = fp2_sub(fp2_sub(m2, x2), y2);
    r0.c0 = fp2_add(x0, fp2_mul_xi(y2));
    r0.c1 = fp2_add(x1, y0);
    r0.c2 = fp2_add(x2, y1);
}
// complex-squaring recombination: r.c0 = s - ab - v ab, r.c1 = 2 ab
ECG_FP6_FN void fp12_sqr_combine(Fp12& r, const Fp6& s, const Fp6& ab) {
    const Fp2 x0 = ab.c0, x1 = ab.c1, x2 = ab.c2;
    const Fp2 s0 = s.c0, s1 = s.c1, s2 = s.c2;
    r.c0.c0 = fp2_sub(fp2_sub(s0, x0), fp2_mul_xi(x2));
    r.c0.c1 = fp2_sub(fp2_sub(s1, x1), x0);
    r.c0.c2 = fp2_sub(fp2_sub(s2, x2), x1);
    r.c1.c0 = fp2_dbl(x0);
    r.c1.c1 = fp2_dbl(x1);
    r.c1.c2 = fp2_dbl(x2);
}

ECG_HD void fp6_inv_inl(Fp6& r, const Fp6& a) {
    Fp2 c0 = fp2_sub(fp2_sqrx(a.c0), fp2_mul_xi(fp2_mulx(a.c1, a.c2)));
    Fp2 c1 = fp2_sub(fp2_mul_xi(fp2_sqrx(a.c2)), fp2_mulx(a.c0, a.c1));
    Fp2 c2 = fp2_sub(fp2_sqrx(a.c1), fp2_mulx(a.c0, a.c2));
    Fp2 t = fp2_add(fp2_mulx(a.c0, c0), fp2_mul_xi(fp2_add(fp2_mulx(a.c2, c1), fp2_mulx(a.c1, c2))));
    Fp2 ti = fp2_inv(t);
    r.c0 = fp2_mulx(c0, ti);
    r.c1 = fp2_mulx(c1, ti);
    r.c2 = fp2_mulx(c2, ti);
}
ECG_HD_NOINLINE void fp6_inv(Fp6& r, const Fp6& a) {
    const Fp6 x = ecg_priv_load(a);
    Fp6 z;
    fp6_inv_inl(z, x);
    ecg_priv_store(r, z);
}

// ---------------------------------------------------------------------------------------------
// Fp12
// ---------------------------------------------------------------------------------------------
ECG_HD void fp12_set_one(Fp12& r) {
    r.c0 = fp6_one();
    r.c1 = fp6_zero();
}
ECG_HD bool fp12_is_one(const Fp12& a) {
    return fp_eq(a.c0.c0.c0, fp_one()) && fp_is_zero(a.c0.c0.c1) && fp2_is_zero(a.c0.c1) && fp2_is_zero(a.c0.c2) &&
           fp2_is_zero(a.c1.c0) && fp2_is_zero(a.c1.c1) && fp2_is_zero(a.c1.c2);
}
ECG_HD void fp12_conj(Fp12& r, const Fp12& a) {
    r.c0 = a.c0;
    fp6_neg(r.c1, a.c1);
}
// 3 Fp6 products.  r may alias a or b.
ECG_HD void fp12_mul_inl(Fp12& r, const Fp12& a, const Fp12& b) {
    Fp6 t0, t1, m;
    fp6_mul(t0, a.c0, b.c0);
    fp6_mul(t1, a.c1, b.c1);
    fp6_mul_sums(m, a.c0, a.c1, b.c0, b.c1);
    fp12_karatsuba_combine(r.c0, r.c1, m, t0, t1);
}
// out of line: operands and result are locals of the caller (private segment, see ecg_priv_load)
ECG_FP12_FN void fp12_mul(Fp12& r, const Fp12& a, const Fp12& b) {
    const Fp12 x = ecg_priv_load(a), y = ecg_priv_load(b);
    Fp12 z;
    fp12_mul_inl(z, x, y);
    ecg_priv_store(r, z);
}
// complex squaring, 2 Fp6 products: c0 = (a0 + a1)(a0 + v a1) - a0a1 - v a0a1, c1 = 2 a0a1
ECG_MILLER_DBL_FN void fp12_sqr(Fp12& r, const Fp12& a) {
    Fp6 ab, s;
    fp6_mul(ab, a.c0, a.c1);
    fp6_mul_sqr_sums(s, a.c0, a.c1);
    fp12_sqr_combine(r, s, ab);
}
// f * ((l0 + l1 v) + (l2 v) w): the Miller-loop line shape on the M-twist.  Schoolbook over the sparse operand: with
// f = (a0, a1, a2) + (b0, b1, b2) w every Fp2 coefficient of the product is one sum of three Fp2 products,
//   c0: a0 l0 + xi a2 l1 + xi b1 l2 | a0 l1 + a1 l0 + xi b2 l2 | a1 l1 + a2 l0 + b0 l2
//   c1: xi a2 l2 + b0 l0 + xi b2 l1 | a0 l2 + b0 l1 + b1 l0    | a1 l2 + b1 l1 + b2 l0
// 72 half-products, 12 reductions, no recombination (the Karatsuba form needs 60 + 18 and 20 modular additions).
// Components of f, l1, l2 < 2p, of l0 < K0 p (the doubling step hands in a lazy sum < 6p); xi-multiples < 4p: a sum is
// below 2 * 4p * K0 p + 4 * 4p * 2p <= 80 p^2.
template <int K0>
ECG_MILLER_DBL_FN void fp12_mul_by_line(Fp12& f, const Fp2& l0, const Fp2& l1, const Fp2& l2) {
    static_assert(8 * K0 + 32 < 632, "sum of products would not reduce below 2p");
    const Fp2 a0 = f.c0.c0, a1 = f.c0.c1, a2 = f.c0.c2, b0 = f.c1.c0, b1 = f.c1.c1, b2 = f.c1.c2;
    const Fp2 xa2 = fp2_mul_xi_lazy<2>(a2), xb1 = fp2_mul_xi_lazy<2>(b1), xb2 = fp2_mul_xi_lazy<2>(b2);
    const Fp n0 = fp_neg_lazy<K0>(l0.c1), n1 = fp_neg_lazy<2>(l1.c1), n2 = fp_neg_lazy<2>(l2.c1);
    const Fp2 y[3] = {l0, l1, l2};
    const Fp ny[3] = {n0, n1, n2};
    const Fp2 y120[3] = {l1, l0, l2};
    const Fp ny120[3] = {n1, n0, n2};
    const Fp2 y201[3] = {l2, l0, l1};
    const Fp ny201[3] = {n2, n0, n1};
    const Fp2 y210[3] = {l2, l1, l0};
    const Fp ny210[3] = {n2, n1, n0};
    {
        const Fp2 x[3] = {a0, xa2, xb1};
        f.c0.c0 = fp2_sumprod<3>(x, y, ny);
    }
    {
        const Fp2 x[3] = {a0, a1, xb2};
        f.c0.c1 = fp2_sumprod<3>(x, y120, ny120);
    }
    {
        const Fp2 x[3] = {a1, a2, b0};
        f.c0.c2 = fp2_sumprod<3>(x, y120, ny120);
    }
    {
        const Fp2 x[3] = {xa2, b0, xb2};
        f.c1.c0 = fp2_sumprod<3>(x, y201, ny201);
    }
    {
        const Fp2 x[3] = {a0, b0, b1};
        f.c1.c1 = fp2_sumprod<3>(x, y210, ny210);
    }
    {
        const Fp2 x[3] = {a1, b1, b2};
        f.c1.c2 = fp2_sumprod<3>(x, y210, ny210);
    }
}
ECG_HD void fp12_inv_inl(Fp12& r, const Fp12& a) {
    Fp6 t0, t1;
    fp6_mul(t0, a.c0, a.c0);
    fp6_mul(t1, a.c1, a.c1);
    fp6_mul_v(t1, t1);
    fp6_sub(t0, t0, t1);
    fp6_inv(t1, t0);
    fp6_mul(r.c0, a.c0, t1);
    fp6_mul(t0, a.c1, t1);
    fp6_neg(r.c1, t0);
}
ECG_HD_NOINLINE void fp12_inv(Fp12& r, const Fp12& a) {
    const Fp12 x = ecg_priv_load(a);
    Fp12 z;
    fp12_inv_inl(z, x);
    ecg_priv_store(r, z);
}
// Frobenius a -> a^p: with a = sum_k a_k w^k (c0 = (a0, a2, a4), c1 = (a1, a3, a5)),
// a_k -> conj(a_k) * xi^(k (p-1)/6).
ECG_HD void fp12_frob_inl(Fp12& r, const Fp12& a) {
    r.c0.c0 = fp2_conj(a.c0.c0);
    r.c1.c0 = fp2_mulx(fp2_conj(a.c1.c0), blsc::FROB_GAMMA[1]);
    r.c0.c1 = fp2_mulx(fp2_conj(a.c0.c1), blsc::FROB_GAMMA[2]);
    r.c1.c1 = fp2_mulx(fp2_conj(a.c1.c1), blsc::FROB_GAMMA[3]);
    r.c0.c2 = fp2_mulx(fp2_conj(a.c0.c2), blsc::FROB_GAMMA[4]);
    r.c1.c2 = fp2_mulx(fp2_conj(a.c1.c2), blsc::FROB_GAMMA[5]);
}
ECG_HD_NOINLINE void fp12_frob(Fp12& r, const Fp12& a) {
    const Fp12 x = ecg_priv_load(a);
    Fp12 z;
    fp12_frob_inl(z, x);
    ecg_priv_store(r, z);
}

// Granger-Scott squaring for elements of the cyclotomic subgroup (after the easy part of the
// final exponentiation): 3 Fp4 squarings instead of 12 Fp2 products.
// Fp4 squaring (a + b s)^2, s^2 = xi: c0 = a^2 + xi b^2, c1 = 2ab as four sums of products over lazy operands --
//   c0.re = (ar + ai)(ar - ai) + (br + bi)(br - bi) - 2 br bi      c0.im = 2 ar ai + (br + bi)(br - bi) + 2 br bi
//   c1.re = 2 ar br - 2 ai bi                                      c1.im = 2 ar bi + 2 ai br
// 10 half-products, 4 reductions.  Components of a, b < K p (K = 2 or 4); every lazy operand < 2 K p: sums below
// (4 + 4 + 2) K^2 p^2 = 160 p^2 at K = 4.
template <int K>
ECG_HD void fp4_sqr(Fp2& c0, Fp2& c1, const Fp2& a, const Fp2& b) {
    const Fp sa = fp_add_lazy(a.c0, a.c1), da = fp_sub_lazy_k<K>(a.c0, a.c1);
    const Fp sb = fp_add_lazy(b.c0, b.c1), db = fp_sub_lazy_k<K>(b.c0, b.c1);
    const Fp b2r = fp_add_lazy(b.c0, b.c0), nbi = fp_neg_lazy<K>(b.c1);
    const Fp a2r = fp_add_lazy(a.c0, a.c0), a2i = fp_add_lazy(a.c1, a.c1);
    {
        const Fp x[3] = {sa, sb, b2r}, y[3] = {da, db, nbi};
        c0.c0 = fp_sumprod<3>(x, y);
    }
    {
        const Fp x[3] = {a2r, sb, b2r}, y[3] = {a.c1, db, b.c1};
        c0.c1 = fp_sumprod<3>(x, y);
    }
    c1.c0 = fp_sumprod2(a2r, b.c0, a2i, nbi);
    c1.c1 = fp_sumprod2(a2r, b.c1, a2i, b.c0);
}
// The linear step z' = 3t +- 2z of every coefficient (t: the Fp4 squaring's output, < 2p) in one pass and two conditional
// subtractions (fp_gs_lin) instead of three modular operations: a RUN of squarings keeps its coefficients below 4p -- which the
// Fp4 squaring's lazy operands absorb -- and whoever ends the run brings them below 2p (fp12_cyc_normalize).  Round 3: the
// linear steps were 26 % of a squaring's instructions.
template <int S, int KT = 2>
ECG_HD Fp2 fp2_gs_lin(const Fp2& t, const Fp2& z) {
    return Fp2{fp_gs_lin<S, KT, 4, 4>(t.c0, z.c0), fp_gs_lin<S, KT, 4, 4>(t.c1, z.c1)};
}
ECG_HD Fp2 fp2_below_2p(const Fp2& a) { return Fp2{fp_cond_sub(a.c0, blsc::P2), fp_cond_sub(a.c1, blsc::P2)}; }  // a < 4p
// coefficients < 4p in, < 4p out
ECG_HD void fp12_cyclotomic_sqr_run(Fp12& r, const Fp12& f) {
    Fp2 z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
    Fp2 t0, t1, t2, t3;
    fp4_sqr<4>(t0, t1, z0, z1);
    z0 = fp2_gs_lin<-1>(t0, z0);
    z1 = fp2_gs_lin<+1>(t1, z1);
    fp4_sqr<4>(t0, t1, z2, z3);
    fp4_sqr<4>(t2, t3, z4, z5);
    z4 = fp2_gs_lin<-1>(t0, z4);
    z5 = fp2_gs_lin<+1>(t1, z5);
    z2 = fp2_gs_lin<+1, 4>(fp2_mul_xi_lazy<2>(t3), z2);  // xi t3: components < 4p
    z3 = fp2_gs_lin<-1>(t2, z3);
    r.c0.c0 = z0;
    r.c0.c1 = z4;
    r.c0.c2 = z3;
    r.c1.c0 = z2;
    r.c1.c1 = z1;
    r.c1.c2 = z5;
}
ECG_HD void fp12_cyc_normalize(Fp12& a) {  // coefficients < 4p -> < 2p
    a.c0.c0 = fp2_below_2p(a.c0.c0);
    a.c0.c1 = fp2_below_2p(a.c0.c1);
    a.c0.c2 = fp2_below_2p(a.c0.c2);
    a.c1.c0 = fp2_below_2p(a.c1.c0);
    a.c1.c1 = fp2_below_2p(a.c1.c1);
    a.c1.c2 = fp2_below_2p(a.c1.c2);
}
// one squaring, coefficients < 2p in and out
ECG_HD void fp12_cyclotomic_sqr_inl(Fp12& r, const Fp12& f) {
    fp12_cyclotomic_sqr_run(r, f);
    fp12_cyc_normalize(r);
}
// Karabina's compressed squaring ("Squaring in cyclotomic subgroups", Math. Comp. 2013): the four coefficients z2 .. z5 of an
// element of the cyclotomic subgroup determine the other two, and squaring THEM is two of the three Fp4 squarings above --
// with the coefficient naming of fp12_cyclotomic_sqr_run the update of z2 .. z5 does not read z0, z1 at all.  A run of k
// squarings between two products costs 2k Fp4 squarings + one decompression (an Fp2 inversion, 3 squarings, 4 products)
// instead of 3k: worth it for the runs of 32 and 16 in the exponent |x| (bls_pairing.h fp12_cyc_pow_x).
// Coefficients < 4p in and out (a run; fp2_below_2p before the decompression).
ECG_HD void fp12_cyclotomic_sqr_compressed(Fp2& z2, Fp2& z3, Fp2& z4, Fp2& z5) {
    Fp2 t0, t1, t2, t3;
    fp4_sqr<4>(t0, t1, z2, z3);
    fp4_sqr<4>(t2, t3, z4, z5);
    z4 = fp2_gs_lin<-1>(t0, z4);
    z5 = fp2_gs_lin<+1>(t1, z5);
    z2 = fp2_gs_lin<+1, 4>(fp2_mul_xi_lazy<2>(t3), z2);
    z3 = fp2_gs_lin<-1>(t2, z3);
}
// z0, z1 from z2 .. z5:  z1 = (xi z5^2 + 3 z4^2 - 2 z3) / (4 z2),  z0 = xi (2 z1^2 + z2 z5 - 3 z3 z4) + 1;  for z2 = 0:
// z1 = 2 z4 z5 / z3 (the same z0 formula; z3 = 0 as well: the element is 1, and 1 / 0 = 0 here gives exactly that).  Both
// numerators are computed and one is selected per lane: no divergent branch around a product.
ECG_HD Fp2 fp2_select(bool c, const Fp2& a, const Fp2& b) {
    Fp2 r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        r.c0.l[i] = c ? a.c0.l[i] : b.c0.l[i];
        r.c1.l[i] = c ? a.c1.l[i] : b.c1.l[i];
    }
    return r;
}
ECG_HD void fp12_cyclotomic_decompress(Fp2& z0, Fp2& z1, const Fp2& z2, const Fp2& z3, const Fp2& z4, const Fp2& z5) {
    const bool z2_zero = fp2_is_zero(z2);
    const Fp2 s4 = fp2_sqrx(z4);
    const Fp2 n_a = fp2_sub(fp2_add(fp2_mul_xi(fp2_sqrx(z5)), fp2_add(fp2_dbl(s4), s4)), fp2_dbl(z3));
    const Fp2 n_b = fp2_dbl(fp2_mulx(z4, z5));
    const Fp2 den = fp2_select(z2_zero, z3, fp2_dbl(fp2_dbl(z2)));
    z1 = fp2_mulx(fp2_select(z2_zero, n_b, n_a), fp2_inv(den));
    const Fp2 p34 = fp2_mulx(z3, z4);
    const Fp2 u = fp2_sub(fp2_add(fp2_dbl(fp2_sqrx(z1)), fp2_mulx(z2, z5)), fp2_add(fp2_dbl(p34), p34));
    z0 = fp2_add(fp2_mul_xi(u), fp2_one());
}
ECG_HD_NOINLINE void fp12_cyclotomic_sqr(Fp12& r, const Fp12& f) {
    const Fp12 x = ecg_priv_load(f);
    Fp12 z;
    fp12_cyclotomic_sqr_inl(z, x);
    ecg_priv_store(r, z);
}

}  // namespace ecg
