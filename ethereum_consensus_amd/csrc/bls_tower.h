// Fp6 / Fp12 tower arithmetic for the BLS12-381 pairing on gfx950:
//   Fp6 = Fp2[v]/(v^3 - xi), xi = 1 + i;   Fp12 = Fp6[w]/(w^2 - v).
// (blst's fp12_tower.c layer under /root/reference/ethereum-consensus/src/crypto/bls.rs:69-71,
// 102-106, 122-126 -- the verify calls.)  An Fp12 is 12 x 13 dwords: it never fits in VGPRs next
// to its operands, so the Fp6-level routines work memory-to-memory on references (the lane's private
// segment) and keep everything inside them in registers; the Fp6 additions of the Fp12 formulas are
// fused into the routine that consumes or produces their operands, because the lane kernels are bound
// by private-segment traffic to HBM, not by arithmetic.
#pragma once
#include "bls_fp.h"

// Call structure of the lane kernels.  A routine that keeps values live across the out-of-line Fp products holds them
// in callee-saved registers and must save / restore those at its own entry / exit: measured 229-282 dwords per
// Fp6-product call, the bulk of k_pairing's private-segment traffic.  Inlining a level removes its saves (a kernel has
// no caller to save for) at the price of code size:  ECG_INLINE_LEVEL 0: Fp6- and Fp12-level routines are calls (40.9 ms);
// 1 (default): Fp6-level routines inline into the Fp12-level ones (39.3 ms, a third fewer saves) and the doubling iteration
// of the Miller loop (accumulator squaring, doubling step, line multiplication) inlines into miller_loop (36.5 ms);
// inlining the cyclotomic squaring into its loop as well changes nothing; 2: all Fp12-level routines and
// the Miller steps inline as well (47 ms: one 2.8 MB function, the register allocator spills more than it saves).
#ifndef ECG_INLINE_LEVEL
#define ECG_INLINE_LEVEL 1
#endif
#if ECG_INLINE_LEVEL >= 1
#define ECG_FP6_FN ECG_HD
#else
#define ECG_FP6_FN ECG_HD_NOINLINE
#endif
#if ECG_INLINE_LEVEL >= 2
#define ECG_FP12_FN ECG_HD
#else
#define ECG_FP12_FN ECG_HD_NOINLINE
#endif
// the doubling iteration of the Miller loop (accumulator squaring, doubling step, line multiplication): one instance each
#if ECG_INLINE_LEVEL >= 1 || defined(ECG_INLINE_MILLER_DBL)
#define ECG_MILLER_DBL_FN ECG_HD
#else
#define ECG_MILLER_DBL_FN ECG_HD_NOINLINE
#endif
// the cyclotomic squaring inside the 63-step exponentiation loops of the final exponentiation
#if ECG_INLINE_LEVEL >= 2 || defined(ECG_INLINE_CYC_SQR)
#define ECG_CYC_SQR_FN ECG_HD
#else
#define ECG_CYC_SQR_FN ECG_HD_NOINLINE
#endif

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
// Karatsuba core, 6 Fp2 products, operands in registers.  Operand components may be lazy sums < 4p: the inner
// pre-sums are product operands too (< 8p into fp2_mul, which allows 8p).
ECG_HD void fp6_mul_core(Fp6& r, const Fp2& a0, const Fp2& a1, const Fp2& a2, const Fp2& b0, const Fp2& b1, const Fp2& b2) {
    Fp2 t0 = fp2_mulx(a0, b0);
    Fp2 t1 = fp2_mulx(a1, b1);
    Fp2 t2 = fp2_mulx(a2, b2);
    Fp2 m12 = fp2_mulx(fp2_add_lazy(a1, a2), fp2_add_lazy(b1, b2));
    Fp2 m01 = fp2_mulx(fp2_add_lazy(a0, a1), fp2_add_lazy(b0, b1));
    Fp2 m02 = fp2_mulx(fp2_add_lazy(a0, a2), fp2_add_lazy(b0, b2));
    r.c0 = fp2_add(t0, fp2_mul_xi(fp2_sub(fp2_sub(m12, t1), t2)));
    r.c1 = fp2_add(fp2_sub(fp2_sub(m01, t0), t1), fp2_mul_xi(t2));
    r.c2 = fp2_add(fp2_sub(fp2_sub(m02, t0), t2), t1);
}
// r may alias a or b (operands are loaded before the result is stored).
ECG_FP6_FN void fp6_mul(Fp6& r, const Fp6& a, const Fp6& b) {
    const Fp2 a0 = a.c0, a1 = a.c1, a2 = a.c2, b0 = b.c0, b1 = b.c1, b2 = b.c2;
    fp6_mul_core(r, a0, a1, a2, b0, b1, b2);
}
// a * (c0 + c1 v): 5 Fp2 products
ECG_FP6_FN void fp6_mul_by_01(Fp6& r, const Fp6& a, const Fp2& c0, const Fp2& c1) {
    Fp2 t0 = fp2_mulx(a.c0, c0);
    Fp2 t1 = fp2_mulx(a.c1, c1);
    Fp2 mid = fp2_sub(fp2_sub(fp2_mulx(fp2_add_lazy(a.c0, a.c1), fp2_add_lazy(c0, c1)), t0), t1);
    Fp2 s2b = fp2_mulx(a.c2, c1);
    Fp2 s2a = fp2_mulx(a.c2, c0);
    r.c0 = fp2_add(t0, fp2_mul_xi(s2b));
    r.c1 = mid;
    r.c2 = fp2_add(t1, s2a);
}
// a * (c1 v): 3 Fp2 products
ECG_FP6_FN void fp6_mul_by_1(Fp6& r, const Fp6& a, const Fp2& c1) {
    Fp2 t0 = fp2_mul_xi(fp2_mulx(a.c2, c1));
    Fp2 t1 = fp2_mulx(a.c0, c1);
    Fp2 t2 = fp2_mulx(a.c1, c1);
    r.c0 = t0;
    r.c1 = t1;
    r.c2 = t2;
}
// r = (a0 + a1)(b0 + b1): the middle product of the Fp12 Karatsuba, sums formed on the fly.  r must not alias.
ECG_FP6_FN void fp6_mul_sums(Fp6& r, const Fp6& a0, const Fp6& a1, const Fp6& b0, const Fp6& b1) {
    fp6_mul_core(r, fp2_add_lazy(a0.c0, a1.c0), fp2_add_lazy(a0.c1, a1.c1), fp2_add_lazy(a0.c2, a1.c2), fp2_add_lazy(b0.c0, b1.c0),
                 fp2_add_lazy(b0.c1, b1.c1), fp2_add_lazy(b0.c2, b1.c2));
}
// r = (a0 + a1)(a0 + v a1): the first product of the complex squaring.  r must not alias.
ECG_FP6_FN void fp6_mul_sqr_sums(Fp6& r, const Fp6& a0, const Fp6& a1) {
    fp6_mul_core(r, fp2_add_lazy(a0.c0, a1.c0), fp2_add_lazy(a0.c1, a1.c1), fp2_add_lazy(a0.c2, a1.c2),
                 fp2_add_lazy(a0.c0, fp2_mul_xi(a1.c2)), fp2_add_lazy(a0.c1, a1.c0), fp2_add_lazy(a0.c2, a1.c1));
}
// r = (f0 + f1) * (l0 + (l1 + l2) v): the middle product of the sparse line multiplication.  r must not alias.
ECG_FP6_FN void fp6_mul_by_01_sums(Fp6& r, const Fp6& f0, const Fp6& f1, const Fp2& l0, const Fp2& l1, const Fp2& l2) {
    // a_k, c1 < 4p; a0 + a1 < 8p and l0 + c1 < 6p as fp2_mul operands
    const Fp2 a0 = fp2_add_lazy(f0.c0, f1.c0), a1 = fp2_add_lazy(f0.c1, f1.c1), a2 = fp2_add_lazy(f0.c2, f1.c2), c1 = fp2_add_lazy(l1, l2);
    Fp2 t0 = fp2_mulx(a0, l0);
    Fp2 t1 = fp2_mulx(a1, c1);
    Fp2 mid = fp2_sub(fp2_sub(fp2_mulx(fp2_add_lazy(a0, a1), fp2_add_lazy(l0, c1)), t0), t1);
    Fp2 s2b = fp2_mulx(a2, c1);
    Fp2 s2a = fp2_mulx(a2, l0);
    r.c0 = fp2_add(t0, fp2_mul_xi(s2b));
    r.c1 = mid;
    r.c2 = fp2_add(t1, s2a);
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

ECG_HD_NOINLINE void fp6_inv(Fp6& r, const Fp6& a) {
    Fp2 c0 = fp2_sub(fp2_sqrx(a.c0), fp2_mul_xi(fp2_mulx(a.c1, a.c2)));
    Fp2 c1 = fp2_sub(fp2_mul_xi(fp2_sqrx(a.c2)), fp2_mulx(a.c0, a.c1));
    Fp2 c2 = fp2_sub(fp2_sqrx(a.c1), fp2_mulx(a.c0, a.c2));
    Fp2 t = fp2_add(fp2_mulx(a.c0, c0), fp2_mul_xi(fp2_add(fp2_mulx(a.c2, c1), fp2_mulx(a.c1, c2))));
    Fp2 ti = fp2_inv(t);
    r.c0 = fp2_mulx(c0, ti);
    r.c1 = fp2_mulx(c1, ti);
    r.c2 = fp2_mulx(c2, ti);
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
ECG_FP12_FN void fp12_mul(Fp12& r, const Fp12& a, const Fp12& b) {
    Fp6 t0, t1, m;
    fp6_mul(t0, a.c0, b.c0);
    fp6_mul(t1, a.c1, b.c1);
    fp6_mul_sums(m, a.c0, a.c1, b.c0, b.c1);
    fp12_karatsuba_combine(r.c0, r.c1, m, t0, t1);
}
// complex squaring, 2 Fp6 products: c0 = (a0 + a1)(a0 + v a1) - a0a1 - v a0a1, c1 = 2 a0a1
ECG_MILLER_DBL_FN void fp12_sqr(Fp12& r, const Fp12& a) {
    Fp6 ab, s;
    fp6_mul(ab, a.c0, a.c1);
    fp6_mul_sqr_sums(s, a.c0, a.c1);
    fp12_sqr_combine(r, s, ab);
}
// f * ((l0 + l1 v) + (l2 v) w): the Miller-loop line shape on the M-twist, 13 Fp2 products.
ECG_MILLER_DBL_FN void fp12_mul_by_line(Fp12& f, const Fp2& l0, const Fp2& l1, const Fp2& l2) {
    Fp6 aa, bb, m;
    fp6_mul_by_01(aa, f.c0, l0, l1);
    fp6_mul_by_1(bb, f.c1, l2);
    fp6_mul_by_01_sums(m, f.c0, f.c1, l0, l1, l2);
    fp12_karatsuba_combine(f.c0, f.c1, m, aa, bb);
}
ECG_HD_NOINLINE void fp12_inv(Fp12& r, const Fp12& a) {
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
// Frobenius a -> a^p: with a = sum_k a_k w^k (c0 = (a0, a2, a4), c1 = (a1, a3, a5)),
// a_k -> conj(a_k) * xi^(k (p-1)/6).
ECG_HD_NOINLINE void fp12_frob(Fp12& r, const Fp12& a) {
    r.c0.c0 = fp2_conj(a.c0.c0);
    r.c1.c0 = fp2_mulx(fp2_conj(a.c1.c0), blsc::FROB_GAMMA[1]);
    r.c0.c1 = fp2_mulx(fp2_conj(a.c0.c1), blsc::FROB_GAMMA[2]);
    r.c1.c1 = fp2_mulx(fp2_conj(a.c1.c1), blsc::FROB_GAMMA[3]);
    r.c0.c2 = fp2_mulx(fp2_conj(a.c0.c2), blsc::FROB_GAMMA[4]);
    r.c1.c2 = fp2_mulx(fp2_conj(a.c1.c2), blsc::FROB_GAMMA[5]);
}

// Granger-Scott squaring for elements of the cyclotomic subgroup (after the easy part of the
// final exponentiation): 3 Fp4 squarings = 9 Fp2 squarings instead of 12 Fp2 products.
ECG_HD void fp4_sqr(Fp2& c0, Fp2& c1, const Fp2& a, const Fp2& b) {
    Fp2 t0 = fp2_sqrx(a);
    Fp2 t1 = fp2_sqrx(b);
    c0 = fp2_add(fp2_mul_xi(t1), t0);
    c1 = fp2_sub(fp2_sub(fp2_sqrx(fp2_add(a, b)), t0), t1);
}
ECG_CYC_SQR_FN void fp12_cyclotomic_sqr(Fp12& r, const Fp12& f) {
    Fp2 z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
    Fp2 t0, t1, t2, t3;
    fp4_sqr(t0, t1, z0, z1);
    z0 = fp2_sub(t0, z0);
    z0 = fp2_add(fp2_dbl(z0), t0);
    z1 = fp2_add(t1, z1);
    z1 = fp2_add(fp2_dbl(z1), t1);
    fp4_sqr(t0, t1, z2, z3);
    fp4_sqr(t2, t3, z4, z5);
    z4 = fp2_sub(t0, z4);
    z4 = fp2_add(fp2_dbl(z4), t0);
    z5 = fp2_add(t1, z5);
    z5 = fp2_add(fp2_dbl(z5), t1);
    t0 = fp2_mul_xi(t3);
    z2 = fp2_add(t0, z2);
    z2 = fp2_add(fp2_dbl(z2), t0);
    z3 = fp2_sub(t2, z3);
    z3 = fp2_add(fp2_dbl(z3), t2);
    r.c0.c0 = z0;
    r.c0.c1 = z4;
    r.c0.c2 = z3;
    r.c1.c0 = z2;
    r.c1.c1 = z1;
    r.c1.c2 = z5;
}

}  // namespace ecg
