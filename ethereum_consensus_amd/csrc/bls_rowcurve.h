// G2 point arithmetic with one point per 16-lane row (bls_rowfield.h supplies the field interface; the Jacobian routines are
// the generic ones of bls_curve.h, every special case included): the end of the message stage -- the two mapped points added,
// the cofactor cleared (Budroni-Pintore: two multiplications by |x|, 126 doublings and 15 additions), the affine conversion --
// as a kernel for small batches.  hash_to_curve is what blst performs inside every verify call
// (/root/reference/ethereum-consensus/src/crypto/bls.rs:71,126).
#pragma once
#include "bls_rowfield.h"

namespace ecg {

typedef Jac<RFp2> RJ2;

// memory <-> row: limb j of a 13-limb image goes to lane j (the pointer is the same for every lane of the row)
ROW_FN RFp rfp_load(const Fp* src) {
#if defined(__HIPCC__)
    const u32 l = threadIdx.x & 15u;
    return RFp{l < 13 ? src->l[l] : 0u};
#else
    return RFp{row_const_limb(src->l)};
#endif
}
ROW_FN RFp2 rfp2_load(const Fp2* src) { return RFp2{rfp_load(&src->c0), rfp_load(&src->c1)}; }
ROW_FN void rfp_store(Fp* dst, const RFp& a) {  // exact limbs of the representative in [0, p)
    const RowK K = row_k();
    const RFp c = rfp_canon(a, K);
#if defined(__HIPCC__)
    const u32 l = threadIdx.x & 15u;
    if (l < 13) dst->l[l] = c.v;
#else
    for (int l = 0; l < 13; l++) dst->l[l] = c.v.v[l];
#endif
}

// psi(x, y, z) = (conj(x) PSI_X, conj(y) PSI_Y, conj(z))
ROW_FN void r_g2_psi(RJ2& r, const RJ2& p) {
    r.x = f_mul(rfp2_conj(p.x), rfp2_const(blsc::PSI_X));
    r.y = f_mul(rfp2_conj(p.y), rfp2_const(blsc::PSI_Y));
    r.z = rfp2_conj(p.z);
}
// [x^2 - x - 1] P + [x - 1] psi(P) + psi^2(2P), term by term as g2_clear_cofactor (bls_h2c.h)
ECG_HD_NOINLINE void r_g2_clear_cofactor(RJ2& r, const RJ2& p_in) {
    const RJ2 p = p_in;
    RJ2 t1, t2, t3, n;
    jac_mul_xabs(t1, p);
    jac_neg(t1, t1);  // [x] P
    r_g2_psi(t2, p);  // psi(P)
    jac_dbl(t3, p);
    r_g2_psi(t3, t3);
    r_g2_psi(t3, t3);  // psi^2(2P)
    jac_neg(n, t2);
    jac_add(t3, t3, n);   // psi^2(2P) - psi(P)
    jac_add(t2, t1, t2);  // [x] P + psi(P)
    jac_mul_xabs(t2, t2);
    jac_neg(t2, t2);  // [x^2] P + [x] psi(P)
    jac_add(t3, t3, t2);
    jac_neg(n, t1);
    jac_add(t3, t3, n);
    jac_neg(n, p);
    jac_add(t3, t3, n);
    r = t3;
}
// 1 / (a0 + a1 i) = (a0 - a1 i) / (a0^2 + a1^2); the Fp inverse as n^(p - 2) = (n^((p - 3) / 4))^4 n (tab: 16 register images of LDS)
ROW_FN RFp2 rfp2_inv(const RFp2& a, u32* tab) {
    const RowK K = row_k();
    rv32 bv[2][13];
    rfp_spread(bv[0], a.c0);
    rfp_spread(bv[1], a.c1);
    const rv32 sq[2] = {a.c0.v, a.c1.v};
    const RFp n{row_sumprod<2>(sq, bv, K.p)};
    RFp t = rfp_pow_pm3d4(n, tab, K);
    t = rfp_sqr(rfp_sqr(t, K), K);
    const RFp ni = rfp_mul(t, n, K);
    return RFp2{rfp_mul(a.c0, ni, K), rfp_mul(rfp_neg(a.c1, K), ni, K)};
}
// q0 + q1, cofactor, affine: the row's H(m) to memory (pointers uniform over the row)
ROW_FN void r_hash_to_g2_finish(A2* out, const J2* q0, const J2* q1, u32* tab) {
    RJ2 a{rfp2_load(&q0->x), rfp2_load(&q0->y), rfp2_load(&q0->z)};
    const RJ2 b{rfp2_load(&q1->x), rfp2_load(&q1->y), rfp2_load(&q1->z)};
    jac_add(a, a, b);
    r_g2_clear_cofactor(a, a);
    const bool inf = jac_is_inf(a);
    RFp2 x, y;
    f_set_zero(x);
    f_set_zero(y);
    if (!inf) {
        const RFp2 zi = rfp2_inv(a.z, tab);
        const RFp2 zi2 = f_sqr(zi);
        x = f_mul(a.x, zi2);
        y = f_mul(f_mul(a.y, zi2), zi);
    }
    rfp_store(&out->x.c0, x.c0);
    rfp_store(&out->x.c1, x.c1);
    rfp_store(&out->y.c0, y.c0);
    rfp_store(&out->y.c1, y.c1);
#if defined(__HIPCC__)
    if ((threadIdx.x & 15u) == 0) out->inf = inf ? 1u : 0u;
#else
    out->inf = inf ? 1u : 0u;
#endif
}

}  // namespace ecg
