// Fp2 on a ROW PAIR: rows 2m and 2m + 1 of a wave hold the real and the imaginary component of the same value, limb j of the own
// component in lane j.  Linear operations are the row's own (half the work of the one-row form of bls_rowfield.h); a product
// is ONE sum of two products per row -- the real row a0 b0 + (8p - a1) b1, the imaginary row a1 b0 + a0 b1 -- over the partner's
// limbs fetched with one ds_bpermute per operand: 0.55 of the one-row form's instructions per lane.  The same field interface
// f_* as RFp2, so the generic point routines of bls_curve.h run on it unchanged.  Used where the chain is long and the batch
// small: the end of the message stage (126 doublings) and the signature's subgroup check (63).
#pragma once
#include "bls_rowfield.h"

namespace ecg {

struct RP2 {
    rv32 v;  // limb `lane` of this row's component
};
ROW_FN RP2 rp2_const(const Fp2& c) {
    const rv32 c0 = row_const_limb(c.c0.l), c1 = row_const_limb(c.c1.l);
    return RP2{rv_sel(rv_pair_row(), c1, c0)};
}
ROW_FN RP2 f_add(const RP2& a, const RP2& b) { const RowK K = row_k(); return RP2{rfp_add(RFp{a.v}, RFp{b.v}, K).v}; }
ROW_FN RP2 f_sub(const RP2& a, const RP2& b) { const RowK K = row_k(); return RP2{rfp_sub(RFp{a.v}, RFp{b.v}, K).v}; }
ROW_FN RP2 f_dbl(const RP2& a) { return f_add(a, a); }
ROW_FN RP2 f_neg(const RP2& a) { const RowK K = row_k(); return RP2{rfp_neg(RFp{a.v}, K).v}; }
ROW_FN RP2 f_add_lazy(const RP2& a, const RP2& b) { const RowK K = row_k(); return RP2{rfp_add_lazy(RFp{a.v}, RFp{b.v}, K).v}; }
template <int KP>
ROW_FN RP2 f_sub_lazy(const RP2& a, const RP2& b) { const RowK K = row_k(); return RP2{rfp_sub_lazy<KP>(RFp{a.v}, RFp{b.v}, K).v}; }
template <int KP>
ROW_FN RP2 f_neg_lazy(const RP2& a) { const RowK K = row_k(); return RP2{rfp_neg_lazy<KP>(RFp{a.v}, K).v}; }
ROW_FN RP2 f_sub_dbl(const RP2& a, const RP2& b) { const RowK K = row_k(); return RP2{rfp_sub_dbl(RFp{a.v}, RFp{b.v}, K).v}; }
// components < 8p on both sides
ROW_FN RP2 f_mul(const RP2& a, const RP2& b) {
    const RowK K = row_k();
    const rv32 im = rv_pair_row(), a_par = rv_partner(a.v), b_par = rv_partner(b.v);
    rv32 bv[2][13];
    rfp_spread(bv[0], RFp{b.v});
    rfp_spread(bv[1], RFp{b_par});
    const rv32 n_par = rfp_neg_lazy<8>(RFp{a_par}, K).v;
    const rv32 av[2] = {rv_sel(im, a_par, a.v), rv_sel(im, a.v, n_par)};  // real: a0 b0 - a1 b1; imaginary: a0 b1 + a1 b0
    return RP2{row_sumprod<2>(av, bv, K.p)};
}
ROW_FN RP2 f_sqr(const RP2& a) { return f_mul(a, a); }
template <int KP>
ROW_FN RP2 f_sqr_lazy(const RP2& a) { return f_mul(a, a); }
template <int KB0, int KB1>
ROW_FN RP2 f_sp2(const RP2& a0, const RP2& b0, const RP2& a1, const RP2& b1) {
    const RowK K = row_k();
    const rv32 im = rv_pair_row();
    const rv32 a0p = rv_partner(a0.v), a1p = rv_partner(a1.v);
    rv32 bv[4][13];
    rfp_spread(bv[0], RFp{b0.v});
    rfp_spread(bv[1], RFp{rv_partner(b0.v)});
    rfp_spread(bv[2], RFp{b1.v});
    rfp_spread(bv[3], RFp{rv_partner(b1.v)});
    const rv32 n0 = rfp_neg_lazy<8>(RFp{a0p}, K).v, n1 = rfp_neg_lazy<8>(RFp{a1p}, K).v;
    const rv32 av[4] = {rv_sel(im, a0p, a0.v), rv_sel(im, a0.v, n0), rv_sel(im, a1p, a1.v), rv_sel(im, a1.v, n1)};
    return RP2{row_sumprod<4>(av, bv, K.p)};
}
ROW_FN bool f_is_zero(const RP2& a) { const RowK K = row_k(); return !rv_test(rv_pair_any(rfp_canon(RFp{a.v}, K).v)); }
ROW_FN bool f_eq(const RP2& a, const RP2& b) {
    const RowK K = row_k();
    return !rv_test(rv_pair_any(rv_xor(rfp_canon(RFp{a.v}, K).v, rfp_canon(RFp{b.v}, K).v)));
}
ROW_FN void f_set_zero(RP2& a) { a = RP2{rv_splat(0)}; }
ROW_FN void f_set_one(RP2& a) { a = RP2{rv_sel(rv_pair_row(), rv_splat(0), row_const_limb(blsc::ONE.l))}; }
ROW_FN RP2 f_conj(const RP2& a) { const RowK K = row_k(); return RP2{rv_sel(rv_pair_row(), rfp_neg(RFp{a.v}, K).v, a.v)}; }
ROW_FN RP2 f_const2(const RP2*, const Fp2& c) { return rp2_const(c); }
// every limb of a row's value in every lane of the row: the one-lane representation (canonical limbs)
ROW_FN Fp rfp_gather(const RFp& a, const RowK& K) {
    const RFp c = rfp_canon(a, K);
    Fp x;
#if defined(__HIPCC__)
    x.l[0] = rv_bcast<0>(c.v), x.l[1] = rv_bcast<1>(c.v), x.l[2] = rv_bcast<2>(c.v), x.l[3] = rv_bcast<3>(c.v), x.l[4] = rv_bcast<4>(c.v);
    x.l[5] = rv_bcast<5>(c.v), x.l[6] = rv_bcast<6>(c.v), x.l[7] = rv_bcast<7>(c.v), x.l[8] = rv_bcast<8>(c.v), x.l[9] = rv_bcast<9>(c.v);
    x.l[10] = rv_bcast<10>(c.v), x.l[11] = rv_bcast<11>(c.v), x.l[12] = rv_bcast<12>(c.v);
#else
    for (int i = 0; i < 13; i++) x.l[i] = c.v.v[i];
#endif
    return x;
}
// 1 / (a0 + a1 i): the norm on both rows (own square + the partner's); its inverse by the ONE-LANE routine in every lane (fp_inv:
// 30 x 30 division steps, ~19 k instructions -- the exponentiation n^(p - 2) is 460 row products, ~76 k: a row makes a product
// 3 x shorter and this chain 4 x longer), then conj(a) / norm with one component per row.  (`tab` stays in the signature: the
// one-row form of bls_rowcurve.h still exponentiates.)
ROW_FN RP2 f_inv_tab(const RP2& a, u32* tab) {
    (void)tab;
    const RowK K = row_k();
    const RFp sq = rfp_sqr(RFp{a.v}, K);
    const RFp n = rfp_add(sq, RFp{rv_partner(sq.v)}, K);
    const Fp ni = fp_inv(rfp_gather(n, K));
#if defined(__HIPCC__)
    const u32 l = threadIdx.x & 15u;
    u32 v = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) v = l == (u32)i ? ni.l[i] : v;
    const RFp nir{v};
#else
    const RFp nir{row_const_limb(ni.l)};
#endif
    const RFp m = rfp_mul(RFp{a.v}, nir, K);
    return RP2{rv_sel(rv_pair_row(), rfp_neg(m, K).v, m.v)};
}
// memory <-> row pair: each row its component
ROW_FN RP2 f_load2(const RP2*, const Fp2* src) {
#if defined(__HIPCC__)
    const Fp* c = ((threadIdx.x >> 4) & 1u) ? &src->c1 : &src->c0;
    return RP2{rfp_load(c).v};
#else
    return rp2_const(*src);
#endif
}
ROW_FN void f_store2(Fp2* dst, const RP2& a) {
    const RowK K = row_k();
    const RFp c = rfp_canon(RFp{a.v}, K);
#if defined(__HIPCC__)
    const u32 l = threadIdx.x & 15u;
    Fp* d = ((threadIdx.x >> 4) & 1u) ? &dst->c1 : &dst->c0;
    if (l < 13) d->l[l] = c.v;
#else
    for (int l = 0; l < 13; l++) dst->c0.l[l] = c.v.v[l], dst->c1.l[l] = c.v.v[16 + l];
#endif
}
// the lane that writes a point's flag word
ROW_FN bool f_first_lane(const RP2*) {
#if defined(__HIPCC__)
    return (threadIdx.x & 31u) == 0;
#else
    return true;
#endif
}

// ---- Fp on a row as the field of the generic G1 routines (a public key per row: its subgroup check is 126 Fp doublings) ----------
struct R1 {
    rv32 v;
};
ROW_FN R1 f_add(const R1& a, const R1& b) { const RowK K = row_k(); return R1{rfp_add(RFp{a.v}, RFp{b.v}, K).v}; }
ROW_FN R1 f_sub(const R1& a, const R1& b) { const RowK K = row_k(); return R1{rfp_sub(RFp{a.v}, RFp{b.v}, K).v}; }
ROW_FN R1 f_dbl(const R1& a) { return f_add(a, a); }
ROW_FN R1 f_neg(const R1& a) { const RowK K = row_k(); return R1{rfp_neg(RFp{a.v}, K).v}; }
ROW_FN R1 f_add_lazy(const R1& a, const R1& b) { const RowK K = row_k(); return R1{rfp_add_lazy(RFp{a.v}, RFp{b.v}, K).v}; }
template <int KP>
ROW_FN R1 f_sub_lazy(const R1& a, const R1& b) { const RowK K = row_k(); return R1{rfp_sub_lazy<KP>(RFp{a.v}, RFp{b.v}, K).v}; }
template <int KP>
ROW_FN R1 f_neg_lazy(const R1& a) { const RowK K = row_k(); return R1{rfp_neg_lazy<KP>(RFp{a.v}, K).v}; }
ROW_FN R1 f_sub_dbl(const R1& a, const R1& b) { const RowK K = row_k(); return R1{rfp_sub_dbl(RFp{a.v}, RFp{b.v}, K).v}; }
ROW_FN R1 f_mul(const R1& a, const R1& b) { const RowK K = row_k(); return R1{rfp_mul(RFp{a.v}, RFp{b.v}, K).v}; }
ROW_FN R1 f_sqr(const R1& a) { return f_mul(a, a); }
template <int KP>
ROW_FN R1 f_sqr_lazy(const R1& a) { return f_mul(a, a); }
template <int KB0, int KB1>
ROW_FN R1 f_sp2(const R1& a0, const R1& b0, const R1& a1, const R1& b1) {
    const RowK K = row_k();
    rv32 bv[2][13];
    rfp_spread(bv[0], RFp{b0.v});
    rfp_spread(bv[1], RFp{b1.v});
    const rv32 av[2] = {a0.v, a1.v};
    return R1{row_sumprod<2>(av, bv, K.p)};
}
ROW_FN bool f_is_zero(const R1& a) { const RowK K = row_k(); return rfp_is_zero(RFp{a.v}, K); }
ROW_FN bool f_eq(const R1& a, const R1& b) { const RowK K = row_k(); return rfp_eq(RFp{a.v}, RFp{b.v}, K); }
ROW_FN void f_set_zero(R1& a) { a = R1{rv_splat(0)}; }
ROW_FN void f_set_one(R1& a) { a = R1{row_const_limb(blsc::ONE.l)}; }

// ... and the same helpers for the one-row form, so that the routines of bls_rowcurve.h are written once
ROW_FN RFp2 f_conj(const RFp2& a) { return rfp2_conj(a); }
ROW_FN RFp2 f_const2(const RFp2*, const Fp2& c) { return rfp2_const(c); }

}  // namespace ecg
