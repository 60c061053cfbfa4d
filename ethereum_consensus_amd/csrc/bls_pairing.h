// Optimal-ate pairing check on BLS12-381 for the gfx950 BLS path: the e(pk, H(m)) == e(g1, sig)
// equation blst evaluates for /root/reference/ethereum-consensus/src/crypto/bls.rs:71,106,126.
//
// Miller loop over |x| = 0xd201000000010000 on the M-twist with inversion-free Jacobian steps; all
// pairs of one product share the accumulator squaring.  Line through the untwisted T evaluated at
// P = (xP, yP), times w^3 and times an Fp2 factor (both killed by the final exponentiation):
//     doubling:  (E X - 2 Y^2) + (-E Z^2 xP) w^2 + (Z3 Z^2 yP) w^3       E = 3 X^2, Z3 = 2 Y Z
//     addition:  (r xQ - yQ Z3) + (-r xP) w^2 + (Z3 yP) w^3              r = 2 (yQ Z^3 - Y), Z3 = 2 Z H
// i.e. the sparse shape fp12_mul_by_line multiplies by.  Final exponentiation: easy part, then
// 3 (p^4 - p^2 + 1)/r = (x-1)^2 (x+p) (x^2 + p^2 - 1) + 3 with Granger-Scott cyclotomic squarings.
#pragma once
#include "bls_curve.h"

namespace ecg {

struct MillerPair {
    Fp px, py;  // P in E1 affine
    Fp2 qx, qy; // Q in E2 affine
    J2 t;       // running point
    u32 active; // 0: P or Q is infinity, the pair contributes 1
};

ECG_HD void miller_pair_init(MillerPair& m, const A1& p, const A2& q) {
    m.active = (p.inf || q.inf) ? 0u : 1u;
    m.px = p.x;
    m.py = p.y;
    m.qx = q.x;
    m.qy = q.y;
    m.t.x = q.x;
    m.t.y = q.y;
    m.t.z = fp2_one();
}

// T <- 2T, f <- f * line_{T,T}(P)
ECG_MILLER_DBL_FN void miller_dbl_step(Fp12& f, MillerPair& m) {
    const J2& T = m.t;
    Fp2 A = fp2_sqrx(T.x);
    Fp2 B = fp2_sqrx(T.y);
    Fp2 C = fp2_sqrx(B);
    Fp2 D = fp2_dbl(fp2_sub(fp2_sub(fp2_sqrx(fp2_add(T.x, B)), A), C));
    Fp2 E = fp2_mul3(A);
    Fp2 Fq = fp2_sqrx(E);
    Fp2 ZZ = fp2_sqrx(T.z);
    Fp2 Z3 = fp2_dbl(fp2_mulx(T.y, T.z));
    Fp2 l0 = fp2_sub(fp2_mulx(E, T.x), fp2_dbl(B));
    Fp2 l1 = fp2_neg(fp2_mul_fp(fp2_mulx(E, ZZ), m.px));
    Fp2 l2 = fp2_mul_fp(fp2_mulx(Z3, ZZ), m.py);
    Fp2 X3 = fp2_sub(Fq, fp2_dbl(D));
    Fp2 C8 = fp2_dbl(fp2_dbl(fp2_dbl(C)));
    m.t.y = fp2_sub(fp2_mulx(E, fp2_sub(D, X3)), C8);
    m.t.x = X3;
    m.t.z = Z3;
    fp12_mul_by_line(f, l0, l1, l2);
}

// T <- T + Q, f <- f * line_{T,Q}(P)
ECG_HD void miller_add_step_inl(Fp12& f, MillerPair& m) {
    const J2& T = m.t;
    Fp2 Z1Z1 = fp2_sqrx(T.z);
    Fp2 U2 = fp2_mulx(m.qx, Z1Z1);
    Fp2 S2 = fp2_mulx(fp2_mulx(m.qy, T.z), Z1Z1);
    Fp2 H = fp2_sub(U2, T.x);
    Fp2 HH = fp2_sqrx(H);
    Fp2 I = fp2_dbl(fp2_dbl(HH));
    Fp2 J = fp2_mulx(H, I);
    Fp2 rr = fp2_dbl(fp2_sub(S2, T.y));
    Fp2 V = fp2_mulx(T.x, I);
    Fp2 X3 = fp2_sub(fp2_sub(fp2_sqrx(rr), J), fp2_dbl(V));
    Fp2 Y3 = fp2_sub(fp2_mulx(rr, fp2_sub(V, X3)), fp2_dbl(fp2_mulx(T.y, J)));
    Fp2 Z3 = fp2_sub(fp2_sub(fp2_sqrx(fp2_add(T.z, H)), Z1Z1), HH);
    Fp2 l0 = fp2_sub(fp2_mulx(rr, m.qx), fp2_mulx(m.qy, Z3));
    Fp2 l1 = fp2_neg(fp2_mul_fp(rr, m.px));
    Fp2 l2 = fp2_mul_fp(Z3, m.py);
    m.t.x = X3;
    m.t.y = Y3;
    m.t.z = Z3;
    fp12_mul_by_line(f, l0, l1, l2);
}
// out of line (5 of the 68 iterations): accumulator and pair are locals of miller_loop
ECG_FP12_FN void miller_add_step(Fp12& f, MillerPair& m) {
    Fp12 x = ecg_priv_load(f);
    MillerPair y = ecg_priv_load(m);
    miller_add_step_inl(x, y);
    ecg_priv_store(f, x);
    ecg_priv_store(m, y);
}

// f = prod_k f_{|x|,Q_k}(P_k), conjugated (x < 0)
// The accumulator is a local VALUE whose address never leaves this function (the five addition steps work on a copy):
// it lives in VGPRs/AGPRs across the whole doubling iteration instead of making three round trips through the
// private segment per bit.
ECG_HD_NOINLINE void miller_loop(Fp12& f, MillerPair* pairs, int n) {
    bool any = false;
    MillerPair lp[2];  // n <= 2 (verify: 2 pairs; aggregate_verify: 1 per lane); private copies, see ecg_priv_load
    for (int k = 0; k < n && k < 2; k++) {
        lp[k] = ecg_priv_load(pairs[k]);
        any = any || lp[k].active;
    }
    Fp12 acc;
    fp12_set_one(acc);
    if (any) {
        for (int b = 62; b >= 0; b--) {
            if (b != 62) fp12_sqr(acc, acc);
            for (int k = 0; k < n; k++)
                if (lp[k].active) miller_dbl_step(acc, lp[k]);
            if ((blsc::X_ABS >> b) & 1)
                for (int k = 0; k < n; k++)
                    if (lp[k].active) {
                        Fp12 t = acc;
                        miller_add_step(t, lp[k]);
                        acc = t;
                    }
        }
        fp12_conj(acc, acc);
    }
    ecg_priv_store(f, acc);
}

// a^x for a in the cyclotomic subgroup (x < 0: conjugate)
ECG_HD_NOINLINE void fp12_cyc_pow_x(Fp12& r, const Fp12& a) {
    // The running value is register-resident (the squaring is inlined, the five products work on a copy); the BASE is not: it
    // is needed five times in 63 iterations, and 156 more live dwords under the squaring cost 64 reloads per iteration.  `a`
    // may alias `r` (written only at the end); it is copied once so that the callee of the product never sees `r`'s storage.
    Fp12 base_mem;
    Fp12 acc = ecg_priv_load(a);
    ecg_priv_store(base_mem, acc);
    for (int b = 62; b >= 0; b--) {
        fp12_cyclotomic_sqr_inl(acc, acc);
        if ((blsc::X_ABS >> b) & 1) {
            // (the product inlined here as well -- running value never leaves the registers, 5 k fewer private-segment
            // instructions per pairing -- builds a kernel that does not terminate on the device, while the host build of the same
            // source passes every test: not pursued, the out-of-line product stays)
            Fp12 t = acc;
            fp12_mul(t, t, base_mem);
            acc = t;
        }
    }
    fp12_conj(acc, acc);
    ecg_priv_store(r, acc);
}

// f^(3 (p^12 - 1)/r); the factor 3 is coprime to r so "== 1" is unaffected
ECG_HD_NOINLINE void final_exponentiation(Fp12& r, const Fp12& f) {
    Fp12 t, u, a, b, c;
    const Fp12 f0 = ecg_priv_load(f);
    // easy part: (p^6 - 1)(p^2 + 1)
    fp12_conj(t, f0);
    fp12_inv(u, f0);
    fp12_mul(t, t, u);
    fp12_frob(u, t);
    fp12_frob(u, u);
    fp12_mul(t, u, t);
    // hard part
    fp12_cyc_pow_x(a, t);
    fp12_conj(u, t);
    fp12_mul(a, a, u);  // t^(x-1)
    fp12_cyc_pow_x(b, a);
    fp12_conj(u, a);
    fp12_mul(a, b, u);  // t^((x-1)^2)
    fp12_cyc_pow_x(b, a);
    fp12_frob(u, a);
    fp12_mul(b, b, u);  // a^(x+p)
    fp12_cyc_pow_x(c, b);
    fp12_cyc_pow_x(c, c);
    fp12_frob(u, b);
    fp12_frob(u, u);
    fp12_mul(c, c, u);
    fp12_conj(u, b);
    fp12_mul(c, c, u);  // b^(x^2 + p^2 - 1)
    fp12_cyclotomic_sqr(u, t);
    fp12_mul(u, u, t);  // t^3
    fp12_mul(c, c, u);
    ecg_priv_store(r, c);
}

// e(p0, q0) * e(p1, q1) == 1 ?
ECG_HD bool pairing_product2_is_one(const A1& p0, const A2& q0, const A1& p1, const A2& q1) {
    MillerPair pr[2];
    miller_pair_init(pr[0], p0, q0);
    miller_pair_init(pr[1], p1, q1);
    Fp12 f, e;
    miller_loop(f, pr, 2);
    final_exponentiation(e, f);
    return fp12_is_one(e);
}

}  // namespace ecg
