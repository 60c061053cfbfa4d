// The final exponentiation of the pairing check on TWO lanes per tuple: the other half of bls_pair2.h (lane 2t = the real parts,
// lane 2t + 1 = the imaginary parts of every Fp2 coefficient of tuple t).  Why it exists: the Miller loop on a lane pair is
// ONE wave per SIMD up to half a round of lanes (32 768 tuples) and takes half the one-lane loop's time there, but the one-lane
// final exponentiation behind it is a 10.4 ms chain on half the SIMDs whatever the batch -- this is that chain at ~0.57 of the
// instructions per lane.  (At 65 536 tuples the pair of kernels is two waves per SIMD and does not beat the one-lane kernel:
// DESIGN.md 3.3a has both measurements.)
//
// The arithmetic mirrors bls_pairing.h final_exponentiation / bls_tower.h routine by routine (same sums of products, same lazy
// bounds; each routine names the one it mirrors): easy part (p^6 - 1)(p^2 + 1), hard part (x - 1)^2 (x + p)(x^2 + p^2 - 1) + 3 with
// Granger-Scott squarings, Karabina's compressed squarings on the two long runs of every exponentiation by x.  An Fp4 squaring
// splits like an Fp2 product: each lane computes ONE component of each of the two outputs (a sum of three and a sum of two Fp
// products) from its own and its partner's operand components; the linear steps of the squarings are component-wise and split
// exactly in half.  The base of an exponentiation by x waits in the lane's six LDS slots (312 bytes per lane).
// Replaces, like those, blst's final_exp under /root/reference/ethereum-consensus/src/crypto/bls.rs:71,126.
#pragma once
#include "bls_g2_pair2.h"

namespace ecg {

// ---- Fp12 on lane pairs -------------------------------------------------------------------------------------------------------
ECG_HD void h12_conj(H12& r, const H12& a) {  // fp12_conj
    for (int j = 0; j < 3; j++) r.c[j] = a.c[j];
    for (int j = 3; j < 6; j++) r.c[j] = H2{fp_neg(a.c[j].v)};
}
// fp12_karatsuba_combine: r1 = m - t0 - t1, r0 = t0 + v t1
ECG_HD void h12_karatsuba_combine(H12& r, const H2 (&m)[3], const H2 (&t0)[3], const H2 (&t1)[3]) {
    r.c[3] = h_sub(h_sub(m[0], t0[0]), t1[0]);
    r.c[4] = h_sub(h_sub(m[1], t0[1]), t1[1]);
    r.c[5] = h_sub(h_sub(m[2], t0[2]), t1[2]);
    r.c[0] = h_add(t0[0], h_mul_xi(t1[2]));
    r.c[1] = h_add(t0[1], t1[0]);
    r.c[2] = h_add(t0[2], t1[1]);
}
// fp12_mul_inl: three Fp6 products.  r may alias a or b.
ECG_HD void h12_mul_inl(H12& r, const H12& a, const H12& b) {
    H2 t0[3], t1[3], m[3];
    h6_mul_lazy<2, 2>(t0[0], t0[1], t0[2], a.c[0], a.c[1], a.c[2], b.c[0], b.c[1], b.c[2]);
    h6_mul_lazy<2, 2>(t1[0], t1[1], t1[2], a.c[3], a.c[4], a.c[5], b.c[3], b.c[4], b.c[5]);
    h6_mul_lazy<4, 4>(m[0], m[1], m[2], h_add_lazy(a.c[0], a.c[3]), h_add_lazy(a.c[1], a.c[4]), h_add_lazy(a.c[2], a.c[5]),
                      h_add_lazy(b.c[0], b.c[3]), h_add_lazy(b.c[1], b.c[4]), h_add_lazy(b.c[2], b.c[5]));  // fp6_mul_sums
    h12_karatsuba_combine(r, m, t0, t1);
}
// fp12_mul_by_slots_inl: the second operand is the H12 in this lane's slots 0 .. 5, read where it is used
ECG_HD void h12_mul_by_slots_inl(H12& r, const H12& a) {
    H2 t0[3], t1[3], m[3];
    {
        const H2 b0 = hslot_load(0), b1 = hslot_load(1), b2 = hslot_load(2);
        h6_mul_lazy<2, 2>(t0[0], t0[1], t0[2], a.c[0], a.c[1], a.c[2], b0, b1, b2);
    }
    {
        const H2 b3 = hslot_load(3), b4 = hslot_load(4), b5 = hslot_load(5);
        h6_mul_lazy<2, 2>(t1[0], t1[1], t1[2], a.c[3], a.c[4], a.c[5], b3, b4, b5);
    }
    {
        const H2 s0 = h_add_lazy(hslot_load(0), hslot_load(3)), s1 = h_add_lazy(hslot_load(1), hslot_load(4)),
                 s2 = h_add_lazy(hslot_load(2), hslot_load(5));
        h6_mul_lazy<4, 4>(m[0], m[1], m[2], h_add_lazy(a.c[0], a.c[3]), h_add_lazy(a.c[1], a.c[4]), h_add_lazy(a.c[2], a.c[5]), s0, s1, s2);
    }
    h12_karatsuba_combine(r, m, t0, t1);
}
ECG_HD void hslot_store_h12(const H12& a) {
    for (int j = 0; j < 6; j++) hslot_store(j, a.c[j]);
}
// fp12_mul_slots: r = a * b for two values in the caller's private segment, b through the slots.  r may alias a or b.
ECG_HD_NOINLINE void h12_mul_slots(H12& r, const H12& a, const H12& b) {
    {
        const H12 y = ecg_priv_load(b);
        hslot_store_h12(y);
    }
    H12 x = ecg_priv_load(a);
    h12_mul_by_slots_inl(x, x);
    ecg_priv_store(r, x);
}
// fp12_frob_inl: a_k -> conj(a_k) * xi^(k (p-1)/6); H12 order c0.c0 c0.c1 c0.c2 | c1.c0 c1.c1 c1.c2 = a0 a2 a4 | a1 a3 a5
ECG_HD_NOINLINE void h12_frob(H12& r, const H12& a_in) {
    const H12 a = ecg_priv_load(a_in);
    H12 z;
    z.c[0] = h_conj(a.c[0]);
    z.c[3] = f_mul(h_conj(a.c[3]), h_of(blsc::FROB_GAMMA[1]));
    z.c[1] = f_mul(h_conj(a.c[1]), h_of(blsc::FROB_GAMMA[2]));
    z.c[4] = f_mul(h_conj(a.c[4]), h_of(blsc::FROB_GAMMA[3]));
    z.c[2] = f_mul(h_conj(a.c[2]), h_of(blsc::FROB_GAMMA[4]));
    z.c[5] = f_mul(h_conj(a.c[5]), h_of(blsc::FROB_GAMMA[5]));
    ecg_priv_store(r, z);
}
// fp6_inv_inl
ECG_HD void h6_inv_inl(H2 (&r)[3], const H2& a0, const H2& a1, const H2& a2) {
    const H2 c0 = h_sub(f_sqr(a0), h_mul_xi(f_mul(a1, a2)));
    const H2 c1 = h_sub(h_mul_xi(f_sqr(a2)), f_mul(a0, a1));
    const H2 c2 = h_sub(f_sqr(a1), f_mul(a0, a2));
    const H2 t = h_add(f_mul(a0, c0), h_mul_xi(h_add(f_mul(a2, c1), f_mul(a1, c2))));
    const H2 ti = f_inv(t);
    r[0] = f_mul(c0, ti);
    r[1] = f_mul(c1, ti);
    r[2] = f_mul(c2, ti);
}
// fp12_inv_inl: (a0 - a1 w) / (a0^2 - v a1^2)
ECG_HD_NOINLINE void h12_inv(H12& r, const H12& a_in) {
    const H12 a = ecg_priv_load(a_in);
    H2 t0[3], t1[3], d[3];
    h6_mul_lazy<2, 2>(t0[0], t0[1], t0[2], a.c[0], a.c[1], a.c[2], a.c[0], a.c[1], a.c[2]);
    h6_mul_lazy<2, 2>(t1[0], t1[1], t1[2], a.c[3], a.c[4], a.c[5], a.c[3], a.c[4], a.c[5]);
    // t0 - v t1, v (x0, x1, x2) = (xi x2, x0, x1)
    const H2 e0 = h_sub(t0[0], h_mul_xi(t1[2])), e1 = h_sub(t0[1], t1[0]), e2 = h_sub(t0[2], t1[1]);
    h6_inv_inl(d, e0, e1, e2);
    H12 z;
    h6_mul_lazy<2, 2>(z.c[0], z.c[1], z.c[2], a.c[0], a.c[1], a.c[2], d[0], d[1], d[2]);
    H2 n[3];
    h6_mul_lazy<2, 2>(n[0], n[1], n[2], a.c[3], a.c[4], a.c[5], d[0], d[1], d[2]);
    for (int j = 0; j < 3; j++) z.c[3 + j] = H2{fp_neg(n[j].v)};
    ecg_priv_store(r, z);
}

// ---- cyclotomic squarings ------------------------------------------------------------------------------------------------------
// fp4_sqr<K> on a lane pair: (a + b s)^2 = (a^2 + xi b^2) + (2ab) s, components of a, b < K p.  With (ar, ai), (br, bi) the
// components of a and b, the one-lane routine's four sums are
//   c0.re = (ar + ai)(ar - ai) + (br + bi)(br - bi) + (2 br)(K p - bi)      c0.im = (2 ar) ai + (br + bi)(br - bi) + (2 br) bi
//   c1.re = (2 ar) br + (2 ai)(K p - bi)                                    c1.im = (2 ar) bi + (2 ai) br
// the even lane computes the left column, the odd lane the right one: operands selected per lane, the same two sums of products.
template <int K>
ECG_HD void h4_sqr(H2& c0, H2& c1, const H2& a, const H2& b) {
    const u32 s = h_s();
    const Fp pa = h_xch(a.v), pb = h_xch(b.v);
    const Fp ar = h_sel(s, pa, a.v), ai = h_sel(s, a.v, pa), br = h_sel(s, pb, b.v), bi = h_sel(s, b.v, pb);
    const Fp sb = fp_add_lazy(br, bi), db = fp_sub_lazy_k<K>(br, bi), b2r = fp_add_lazy(br, br), a2r = fp_add_lazy(ar, ar);
    const Fp a2i = fp_add_lazy(ai, ai);
    {
        // even: (ar + ai)(ar - ai + K p); odd: (2 ar) ai
        const Fp x0 = fp_add_lazy(ar, h_sel(s, ar, ai));
        const Fp y0 = h_sel(s, ai, fp_sub_lazy_k<K>(ar, ai));
        const Fp y2 = h_sel(s, bi, fp_neg_lazy<K>(bi));
        const Fp x[3] = {x0, sb, b2r}, y[3] = {y0, db, y2};
        c0 = H2{fp_sumprod<3>(x, y)};
    }
    {
        const Fp y0 = h_sel(s, bi, br);
        const Fp y1 = h_sel(s, br, fp_neg_lazy<K>(bi));
        c1 = H2{fp_sumprod2(a2r, y0, a2i, y1)};
    }
}
template <int S, int KT = 2>
ECG_HD H2 h_gs_lin(const H2& t, const H2& z) {  // fp2_gs_lin: component-wise
    return H2{fp_gs_lin<S, KT, 4, 4>(t.v, z.v)};
}
ECG_HD H2 h_below_2p(const H2& a) { return H2{fp_cond_sub(a.v, blsc::P2)}; }  // a < 4p
// fp12_cyclotomic_sqr_run: coefficients < 4p in, < 4p out.  H12 order: z0 = c[0], z4 = c[1], z3 = c[2], z2 = c[3], z1 = c[4], z5 = c[5]
ECG_HD void h12_cyclotomic_sqr_run(H12& r, const H12& f) {
    H2 z0 = f.c[0], z4 = f.c[1], z3 = f.c[2], z2 = f.c[3], z1 = f.c[4], z5 = f.c[5];
    H2 t0, t1, t2, t3;
    h4_sqr<4>(t0, t1, z0, z1);
    z0 = h_gs_lin<-1>(t0, z0);
    z1 = h_gs_lin<+1>(t1, z1);
    h4_sqr<4>(t0, t1, z2, z3);
    h4_sqr<4>(t2, t3, z4, z5);
    z4 = h_gs_lin<-1>(t0, z4);
    z5 = h_gs_lin<+1>(t1, z5);
    z2 = h_gs_lin<+1, 4>(h_mul_xi_lazy<2>(t3), z2);
    z3 = h_gs_lin<-1>(t2, z3);
    r.c[0] = z0;
    r.c[1] = z4;
    r.c[2] = z3;
    r.c[3] = z2;
    r.c[4] = z1;
    r.c[5] = z5;
}
ECG_HD void h12_cyc_normalize(H12& a) {
    for (int j = 0; j < 6; j++) a.c[j] = h_below_2p(a.c[j]);
}
// fp12_cyclotomic_sqr_compressed
ECG_HD void h12_cyclotomic_sqr_compressed(H2& z2, H2& z3, H2& z4, H2& z5) {
    H2 t0, t1, t2, t3;
    h4_sqr<4>(t0, t1, z2, z3);
    h4_sqr<4>(t2, t3, z4, z5);
    z4 = h_gs_lin<-1>(t0, z4);
    z5 = h_gs_lin<+1>(t1, z5);
    z2 = h_gs_lin<+1, 4>(h_mul_xi_lazy<2>(t3), z2);
    z3 = h_gs_lin<-1>(t2, z3);
}
ECG_HD H2 h2_select(bool c, const H2& a, const H2& b) { return H2{h_sel(c ? 1u : 0u, a.v, b.v)}; }
// fp12_cyclotomic_decompress (the verdict z2 == 0 is the same on both lanes of the pair: f_is_zero)
ECG_HD void h12_cyclotomic_decompress(H2& z0, H2& z1, const H2& z2, const H2& z3, const H2& z4, const H2& z5) {
    const bool z2_zero = f_is_zero(z2);
    const H2 s4 = f_sqr(z4);
    const H2 n_a = h_sub(h_add(h_mul_xi(f_sqr(z5)), h_add(h_dbl(s4), s4)), h_dbl(z3));
    const H2 n_b = h_dbl(f_mul(z4, z5));
    const H2 den = h2_select(z2_zero, z3, h_dbl(h_dbl(z2)));
    z1 = f_mul(h2_select(z2_zero, n_b, n_a), f_inv(den));
    const H2 p34 = f_mul(z3, z4);
    const H2 u = h_sub(h_add(h_dbl(f_sqr(z1)), f_mul(z2, z5)), h_add(h_dbl(p34), p34));
    z0 = h_add(h_mul_xi(u), h_one());
}
ECG_HD_NOINLINE void h12_cyclotomic_sqr(H12& r, const H12& f) {
    H12 x = ecg_priv_load(f);
    h12_cyclotomic_sqr_run(x, x);
    h12_cyc_normalize(x);
    ecg_priv_store(r, x);
}

// fp12_cyc_pow_x: a^x for a in the cyclotomic subgroup (x < 0: conjugate); the base waits in the lane slots
ECG_HD_NOINLINE void h12_cyc_pow_x(H12& r, const H12& a) {
    ECG_LONG_BRANCH_GUARD();
    H12 acc = ecg_priv_load(a);
    hslot_store_h12(acc);
    struct Run {
        u8 n, compressed;
    };
    ECG_CONST Run RUNS[6] = {{1, 0}, {2, 0}, {3, 0}, {9, 0}, {32, 1}, {16, 1}};
    for (int s = 0; s < 6; s++) {
        const u32 n = RUNS[s].n;
        if (RUNS[s].compressed) {
            H2 z2 = acc.c[3], z3 = acc.c[2], z4 = acc.c[1], z5 = acc.c[5];
            for (u32 k = 0; k < n; k++) h12_cyclotomic_sqr_compressed(z2, z3, z4, z5);
            z2 = h_below_2p(z2);
            z3 = h_below_2p(z3);
            z4 = h_below_2p(z4);
            z5 = h_below_2p(z5);
            H2 z0, z1;
            h12_cyclotomic_decompress(z0, z1, z2, z3, z4, z5);
            acc.c[0] = z0;
            acc.c[1] = z4;
            acc.c[2] = z3;
            acc.c[3] = z2;
            acc.c[4] = z1;
            acc.c[5] = z5;
        } else {
            for (u32 k = 0; k < n; k++) h12_cyclotomic_sqr_run(acc, acc);
            h12_cyc_normalize(acc);
        }
        if (s < 5) h12_mul_by_slots_inl(acc, acc);
    }
    h12_conj(acc, acc);
    ecg_priv_store(r, acc);
}

// final_exponentiation: f^(3 (p^12 - 1)/r)
ECG_HD_NOINLINE void h_final_exponentiation(H12& r, const H12& f) {
    H12 t, u, a, b, c;
    const H12 f0 = ecg_priv_load(f);
    // easy part: (p^6 - 1)(p^2 + 1)
    h12_conj(t, f0);
    h12_inv(u, f0);
    h12_mul_slots(t, t, u);
    h12_frob(u, t);
    h12_frob(u, u);
    h12_mul_slots(t, u, t);
    // hard part
    h12_cyc_pow_x(a, t);
    h12_conj(u, t);
    h12_mul_slots(a, a, u);  // t^(x-1)
    h12_cyc_pow_x(b, a);
    h12_conj(u, a);
    h12_mul_slots(a, b, u);  // t^((x-1)^2)
    h12_cyc_pow_x(b, a);
    h12_frob(u, a);
    h12_mul_slots(b, b, u);  // a^(x+p)
    h12_cyc_pow_x(c, b);
    h12_cyc_pow_x(c, c);
    h12_frob(u, b);
    h12_frob(u, u);
    h12_mul_slots(c, c, u);
    h12_conj(u, b);
    h12_mul_slots(c, c, u);  // b^(x^2 + p^2 - 1)
    h12_cyclotomic_sqr(u, t);
    h12_mul_slots(u, u, t);  // t^3
    h12_mul_slots(c, c, u);
    ecg_priv_store(r, c);
}

// fp12_is_one: the same verdict on both lanes of the pair
ECG_HD bool h12_is_one(const H12& a) {
    const u32 s = h_s();
    u32 ok = s ? (fp_is_zero(a.c[0].v) ? 1u : 0u) : (fp_eq(a.c[0].v, fp_one()) ? 1u : 0u);
    for (int j = 1; j < 6; j++) ok &= fp_is_zero(a.c[j].v) ? 1u : 0u;
    return (ok & h_xch_u32(ok)) != 0;
}
// this lane's components of an Fp12 in memory (coefficient order of struct Fp12)
ECG_HD void h12_load(H12& f, const Fp12* in) {
    const Fp* o = reinterpret_cast<const Fp*>(in);
    const u32 s = h_s();
    for (int j = 0; j < 6; j++) f.c[j] = H2{o[2 * j + s]};
}

}  // namespace ecg
