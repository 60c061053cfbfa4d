// G2 point arithmetic on TWO lanes per point (lane 2t: the real parts, lane 2t + 1: the imaginary parts of every Fp2
// coordinate; bls_pair2.h has the exchange primitives) -- for the LATENCY-bound small batches: a lone aggregate's message stage
// ends in one lane adding two points and clearing the cofactor, a 3.5 ms dependent chain of 126 doublings and 15 additions
// (k_h2c_finish; the floor of a slot's sync aggregate, of a block's collector flush, of every committee batch below a few
// thousand tuples).  Half the components per lane is half the instructions per lane, and a batch that small leaves the lanes
// idle anyway.  (For chip-filling batches the same split loses: it adds ~13 % instructions and a second wave per SIMD buys
// less than that)
// The generic point routines of bls_curve.h (jac_dbl_inl, jac_add_inl, jac_mul_xabs, jac_to_aff: templates over the field
// interface f_*) are instantiated with F = H2: this file only supplies that interface.
#pragma once
#include "bls_pair2.h"

namespace ecg {

#if defined(__HIP_DEVICE_COMPILE__)
ECG_D u32 h_xch_u32(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, false); }
#else
inline u32 h_xch_u32(u32 v) {
    Fp e = fp_zero();
    e.l[0] = v;
    return h_xch(e).l[0];
}
#endif

// ---- the field interface of bls_curve.h for F = H2 ----------------------------------------------------------------------------
ECG_HD H2 f_add(const H2& a, const H2& b) { return h_add(a, b); }
ECG_HD H2 f_sub(const H2& a, const H2& b) { return h_sub(a, b); }
ECG_HD H2 f_dbl(const H2& a) { return h_dbl(a); }
ECG_HD H2 f_neg(const H2& a) { return H2{fp_neg(a.v)}; }
ECG_HD H2 f_mul(const H2& a, const H2& b) { return h_mul<8>(a, b); }  // fp2_mul: operand components < 8p
ECG_HD H2 f_sqr(const H2& a) { return h_sqr(a); }
ECG_HD bool f_is_zero(const H2& a) {  // both components: the same verdict on both lanes of the pair
    const u32 z = fp_is_zero(a.v) ? 1u : 0u;
    return (z & h_xch_u32(z)) != 0;
}
ECG_HD void f_set_zero(H2& a) { a = H2{fp_zero()}; }
ECG_HD void f_set_one(H2& a) { a = h_one(); }
ECG_HD H2 f_add_lazy(const H2& a, const H2& b) { return h_add_lazy(a, b); }
template <int K>
ECG_HD H2 f_sub_lazy(const H2& a, const H2& b) { return h_sub_lazy<K>(a, b); }
template <int K>
ECG_HD H2 f_neg_lazy(const H2& a) { return h_neg_lazy<K>(a); }
template <int K>
ECG_HD H2 f_sqr_lazy(const H2& a) { return h_sqr_lazy<K>(a); }
ECG_HD H2 f_sub_dbl(const H2& a, const H2& b) { return H2{fp_sub_dbl(a.v, b.v)}; }
template <int KB0, int KB1>
ECG_HD H2 f_sp2(const H2& a0, const H2& b0, const H2& a1, const H2& b1) {
    const HX x[2] = {h_x(a0), h_x(a1)};
    const HY y[2] = {h_y<KB0>(b0), h_y<KB1>(b1)};
    return h_sumprod<2>(x, y);
}
// 1 / a: d = 1 / (a0^2 + a1^2) on both lanes (the same value: the sum is symmetric), then a0 d | -(a1 d)
ECG_HD H2 f_inv(const H2& a) {
    const Fp sq = fp_sqr(a.v);
    const Fp d = fp_inv(fp_add(sq, h_xch(sq)));
    const Fp m = fp_mul(a.v, d);
    return H2{h_sel(h_s(), fp_neg(m), m)};
}
ECG_HD H2 h_conj(const H2& a) { return H2{h_sel(h_s(), fp_neg(a.v), a.v)}; }
ECG_HD H2 h_of(const Fp2& c) { return H2{h_sel(h_s(), c.c1, c.c0)}; }  // this lane's component of a value both lanes hold

typedef Jac<H2> HJ;

// g2_psi (bls_curve.h)
ECG_HD void h_g2_psi(HJ& r, const HJ& p) {
    r.x = f_mul(h_conj(p.x), h_of(blsc::PSI_X));
    r.y = f_mul(h_conj(p.y), h_of(blsc::PSI_Y));
    r.z = h_conj(p.z);
}
// g2_clear_cofactor (bls_h2c.h), term by term: [x^2 - x - 1] P + [x - 1] psi(P) + psi^2(2P)
ECG_HD_NOINLINE void h_g2_clear_cofactor(HJ& r, const HJ& p_in) {
    const HJ p = ecg_priv_load(p_in);
    HJ t1, t2, t3, n;
    jac_mul_xabs(t1, p);
    jac_neg(t1, t1);  // [x] P
    h_g2_psi(t2, p);  // psi(P)
    jac_dbl(t3, p);
    h_g2_psi(t3, t3);
    h_g2_psi(t3, t3);  // psi^2(2P)
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
    ecg_priv_store(r, t3);
}
// hash_to_g2_finish on a lane pair: q0 + q1, cofactor, affine; the result's components go to `r` in memory
ECG_HD_NOINLINE void h_hash_to_g2_finish(A2* r, const J2& q0_in, const J2& q1_in) {
    const u32 s = h_s();
    const J2 a = q0_in, b = q1_in;  // global memory: both lanes read both points, each keeps its components
    HJ q0{h_of(a.x), h_of(a.y), h_of(a.z)}, q1{h_of(b.x), h_of(b.y), h_of(b.z)};
    jac_add(q0, q0, q1);
    h_g2_clear_cofactor(q0, q0);
    Fp* out = reinterpret_cast<Fp*>(r);  // A2 = {Fp2 x, Fp2 y, u32 inf}: x.c0 x.c1 y.c0 y.c1
    if (jac_is_inf(q0)) {
        out[s] = fp_zero();
        out[2 + s] = fp_zero();
        if (s == 0) r->inf = 1;
        return;
    }
    const H2 zi = f_inv(q0.z);
    const H2 zi2 = f_sqr(zi);
    out[s] = f_mul(q0.x, zi2).v;
    out[2 + s] = f_mul(f_mul(q0.y, zi2), zi).v;
    if (s == 0) r->inf = 0;
}

}  // namespace ecg
