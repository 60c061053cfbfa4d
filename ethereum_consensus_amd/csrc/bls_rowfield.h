// Fp and Fp2 arithmetic with ONE ELEMENT PER 16-LANE ROW (limb j in lane j; an Fp2 value keeps both components in the same
// row, two dwords per lane), supplying the field interface f_* of bls_curve.h -- so the generic point routines (jac_dbl_inl,
// jac_add_inl, jac_mul_xabs, jac_to_aff: every special case of the one-lane code) run on rows unchanged.  This is the latency
// form of the side stages of a small batch: the end of the message stage (addition of the two mapped points, cofactor
// clearing, affine conversion) is a dependent chain of 126 doublings and 15 additions -- 3.5 ms on one lane, 1.6 ms on a lane
// pair (bls_g2_pair2.h), ~0.6 ms on a row.  Same lane-vector vocabulary as bls_row.h: the host simulator runs
// these routines too.
//
// Representation: limbs <= 2^30 + a few (lazy: products accept it), value bounded as the one-lane code bounds it (results of
// products and of the strict operations < 2p; f_*_lazy results up to their K p).  Where the one-lane code ripples a borrow
// through 13 limbs, a row resolves SIGNED limbs in two passes with a bias that cancels (r_norm_signed); a comparison is the sign
// of the top limb of a difference (lane 12, broadcast); exact limbs (for equality tests, sign bits, stores) cost one ballot
// addition (rv_carry_exact).
#pragma once
#include "bls_curve.h"
#include "bls_row.h"

namespace ecg {

struct RFp {
    rv32 v;
};
struct RFp2 {
    RFp c0, c1;
};

// per-lane constants of a row
struct RowK {
    rv32 lane, p, one, mask;
};
ROW_FN rv32 row_const_limb(const u32* limbs13) {  // limb `lane` of a constant (0 beyond limb 12)
#if defined(__HIPCC__)
    const u32 l = threadIdx.x & 15u;
    return l < 13 ? limbs13[l] : 0u;
#else
    rv32 r;
    ROW_EACH r.v[l_] = (l_ & 15) < 13 ? limbs13[l_ & 15] : 0u;
    return r;
#endif
}
ROW_FN RowK row_k() { return RowK{rv_lane(), row_const_limb(blsc::P), row_const_limb(blsc::ONE.l), rv_splat(FP_MASK)}; }
ROW_FN RFp rfp_const(const Fp& c) { return RFp{row_const_limb(c.l)}; }
ROW_FN RFp2 rfp2_const(const Fp2& c) { return RFp2{rfp_const(c.c0), rfp_const(c.c1)}; }

// signed limbs |t_j| < 2^40 of a value that is >= 0 as a whole (or whose sign the caller reads off the top limb) -> exact limbs
// in [0, 2^30) below the top, the top limb (lane 12) keeping the rest as a signed dword: >= 0 exactly when the value is
ROW_FN rv32 r_norm_signed(rv64 t, const RowK& K) {
    const rv32 below_top = rv_lt(K.lane, 12), has_lower = rv_lt(rv_sub(K.lane, rv_splat(1)), 12);  // lanes 0 .. 11 / 1 .. 12
    t = rv_sel64(below_top, rv_add64c(t, 1ull << 40), t);
    t = rv_sel64(has_lower, rv_add64c(t, (u64)0 - (1ull << 10)), t);
    const rv32 mask = rv_sel(below_top, K.mask, rv_splat(0xffffffffu)), zero = rv_splat(0);
    rv32 hi = rv_sel(below_top, rv_lo(rv_sar64(t, 30)), zero);
    rv32 v = rv_add(rv_and(rv_lo(t), mask), rv_from_prev(hi));
    hi = rv_sel(below_top, rv_shr(v, 30), zero);
    // ... and exact below the top: with a limb == 2^30 left standing a small positive value can come out as (.., 2^30, .., top = -1)
    return rv_carry_exact(rv_add(rv_and(v, mask), rv_from_prev(hi)));
}
// c_a a + c_b b + k p, small signed coefficients
ROW_FN rv32 r_lin(rv32 a, int ca, rv32 b, int cb, int k, const RowK& K) {
    rv64 t = rv_mad64s(rv_splat((u32)ca), a, rv_zero64());
    t = rv_mad64s(rv_splat((u32)cb), b, t);
    t = rv_mad64s(rv_splat((u32)k), K.p, t);
    return r_norm_signed(t, K);
}
// x if x < k p else x - k p, for 0 <= x: the sign of x - k p is the sign of its top limb (the limbs below are exact)
ROW_FN rv32 r_cond_sub(rv32 x, int k, const RowK& K) {
    const rv32 d = r_lin(x, 1, x, 0, -k, K);
    const rv32 neg = rv_shr(rv_bcast<12>(d), 31);
    return rv_sel(neg, x, d);
}

// ---- Fp on a row -----------------------------------------------------------------------------------------------------------------
ROW_FN RFp rfp_add_lazy(const RFp& a, const RFp& b, const RowK& K) {  // limbs stay <= 2^30 + 2
    const rv32 s = rv_add(a.v, b.v);
    return RFp{rv_add(rv_and(s, K.mask), rv_from_prev(rv_shr(s, 30)))};
}
template <int KP>
ROW_FN RFp rfp_neg_lazy(const RFp& a, const RowK& K) { return RFp{r_lin(a.v, -1, a.v, 0, KP, K)}; }  // K p - a for a < K p
template <int KP>
ROW_FN RFp rfp_sub_lazy(const RFp& a, const RFp& b, const RowK& K) { return RFp{r_lin(a.v, 1, b.v, -1, KP, K)}; }  // a - b + K p, b < K p
ROW_FN RFp rfp_add(const RFp& a, const RFp& b, const RowK& K) { return RFp{r_cond_sub(rfp_add_lazy(a, b, K).v, 2, K)}; }  // < 2p
ROW_FN RFp rfp_sub(const RFp& a, const RFp& b, const RowK& K) { return RFp{r_cond_sub(r_lin(a.v, 1, b.v, -1, 2, K), 2, K)}; }
ROW_FN RFp rfp_neg(const RFp& a, const RowK& K) { return RFp{r_lin(a.v, -1, a.v, 0, 2, K)}; }  // 2p - a in (0, 2p]
ROW_FN RFp rfp_dbl(const RFp& a, const RowK& K) { return rfp_add(a, a, K); }
// a - 2b in [0, 2p) for a, b < 2p
ROW_FN RFp rfp_sub_dbl(const RFp& a, const RFp& b, const RowK& K) {
    const rv32 t = r_lin(a.v, 1, b.v, -2, 4, K);  // in (0, 6p)
    return RFp{r_cond_sub(r_cond_sub(t, 4, K), 2, K)};
}
// the 13 limbs of b in every lane of the row
ROW_FN void rfp_spread(rv32 (&out)[13], const RFp& b) {
    out[0] = rv_bcast<0>(b.v), out[1] = rv_bcast<1>(b.v), out[2] = rv_bcast<2>(b.v), out[3] = rv_bcast<3>(b.v);
    out[4] = rv_bcast<4>(b.v), out[5] = rv_bcast<5>(b.v), out[6] = rv_bcast<6>(b.v), out[7] = rv_bcast<7>(b.v);
    out[8] = rv_bcast<8>(b.v), out[9] = rv_bcast<9>(b.v), out[10] = rv_bcast<10>(b.v), out[11] = rv_bcast<11>(b.v);
    out[12] = rv_bcast<12>(b.v);
}
ROW_FN RFp rfp_mul(const RFp& a, const RFp& b, const RowK& K) {
    rv32 av[1] = {a.v}, bv[1][13];
    rfp_spread(bv[0], b);
    return RFp{row_sumprod<1>(av, bv, K.p)};
}
ROW_FN RFp rfp_sqr(const RFp& a, const RowK& K) { return rfp_mul(a, a, K); }
// the unique limbs of the unique representative in [0, p) (input <= 2p)
ROW_FN RFp rfp_canon(const RFp& a, const RowK& K) {
    rv32 x = r_cond_sub(rv_carry_exact(a.v), 1, K);
    return RFp{r_cond_sub(x, 1, K)};
}
ROW_FN bool rfp_is_zero(const RFp& a, const RowK& K) { return !rv_test(rv_row_any(rfp_canon(a, K).v)); }
ROW_FN bool rfp_eq(const RFp& a, const RFp& b, const RowK& K) { return !rv_test(rv_row_any(rv_xor(rfp_canon(a, K).v, rfp_canon(b, K).v))); }
ROW_FN RFp rfp_zero() { return RFp{rv_splat(0)}; }

// a^((p - 3) / 4) by the sliding-window schedule of the one-lane code (blsc::POW_PM3D4_SCHED: odd powers a .. a^31 in `tab`,
// 16 register images of LDS owned by the row), 375 squarings + 81 products
ROW_FN RFp rfp_pow_pm3d4(const RFp& a, u32* tab, const RowK& K) {
    const RFp a2 = rfp_sqr(a, K);
    RFp t = a;
    rv_lds_write(tab, K.lane, t.v, rv_splat(1));
    for (int i = 1; i < 16; i++) {
        t = rfp_mul(t, a2, K);
        rv_lds_write(tab, rv_add(rv_splat(16 * i), K.lane), t.v, rv_splat(1));
    }
    RFp acc = rfp_zero();
    bool first = true;
    for (int s = 0; s < blsc::POW_PM3D4_STEPS; s++) {
        const u32 nsq = blsc::POW_PM3D4_SCHED[s][0], idx = blsc::POW_PM3D4_SCHED[s][1];
        if (first) {  // the schedule starts with a table entry (its squaring count is 0 bits of a zero accumulator)
            acc = RFp{rv_lds_read(tab, rv_add(rv_splat(16 * idx), K.lane))};
            first = false;
            continue;
        }
        for (u32 q = 0; q < nsq; q++) acc = rfp_sqr(acc, K);
        if (idx != 255) {
            rv32 av[1] = {acc.v}, bv[1][13];
            rv32 qd[16];
            for (int c = 0; c < 4; c++) rv_lds_read4(tab, rv_splat(16 * idx + 4 * c), qd + 4 * c);
            for (int i = 0; i < 13; i++) bv[0][i] = qd[i];
            acc = RFp{row_sumprod<1>(av, bv, K.p)};
        }
    }
    return acc;
}

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

// ---- Fp2 on a row: the field interface of bls_curve.h ---------------------------------------------------------------------------
// (the constants of a row travel in a thread-local of the kernel: the generic routines call f_*(a, b) without a context)
#if defined(__HIPCC__)
#define ROW_K() row_k()
#else
#define ROW_K() row_k()
#endif
ROW_FN RFp2 f_add(const RFp2& a, const RFp2& b) { const RowK K = ROW_K(); return RFp2{rfp_add(a.c0, b.c0, K), rfp_add(a.c1, b.c1, K)}; }
ROW_FN RFp2 f_sub(const RFp2& a, const RFp2& b) { const RowK K = ROW_K(); return RFp2{rfp_sub(a.c0, b.c0, K), rfp_sub(a.c1, b.c1, K)}; }
ROW_FN RFp2 f_dbl(const RFp2& a) { return f_add(a, a); }
ROW_FN RFp2 f_neg(const RFp2& a) { const RowK K = ROW_K(); return RFp2{rfp_neg(a.c0, K), rfp_neg(a.c1, K)}; }
ROW_FN RFp2 f_add_lazy(const RFp2& a, const RFp2& b) { const RowK K = ROW_K(); return RFp2{rfp_add_lazy(a.c0, b.c0, K), rfp_add_lazy(a.c1, b.c1, K)}; }
template <int KP>
ROW_FN RFp2 f_sub_lazy(const RFp2& a, const RFp2& b) { const RowK K = ROW_K(); return RFp2{rfp_sub_lazy<KP>(a.c0, b.c0, K), rfp_sub_lazy<KP>(a.c1, b.c1, K)}; }
template <int KP>
ROW_FN RFp2 f_neg_lazy(const RFp2& a) { const RowK K = ROW_K(); return RFp2{rfp_neg_lazy<KP>(a.c0, K), rfp_neg_lazy<KP>(a.c1, K)}; }
ROW_FN RFp2 f_sub_dbl(const RFp2& a, const RFp2& b) { const RowK K = ROW_K(); return RFp2{rfp_sub_dbl(a.c0, b.c0, K), rfp_sub_dbl(a.c1, b.c1, K)}; }
// (a0 + a1 i)(b0 + b1 i): two sums of two products; the sign goes to the a-side (a lane's own limb: nothing to broadcast).
// Components < 8p on both sides (the one-lane fp2_mul's contract).
ROW_FN RFp2 f_mul(const RFp2& a, const RFp2& b) {
    const RowK K = ROW_K();
    rv32 bv[2][13];
    rfp_spread(bv[0], b.c0);
    rfp_spread(bv[1], b.c1);
    const rv32 re[2] = {a.c0.v, rfp_neg_lazy<8>(a.c1, K).v};
    const rv32 c0 = row_sumprod<2>(re, bv, K.p);
    rv32 bw[2][13];
    for (int i = 0; i < 13; i++) bw[0][i] = bv[1][i], bw[1][i] = bv[0][i];
    const rv32 im[2] = {a.c0.v, a.c1.v};
    return RFp2{RFp{c0}, RFp{row_sumprod<2>(im, bw, K.p)}};
}
ROW_FN RFp2 f_sqr(const RFp2& a) { return f_mul(a, a); }
template <int KP>
ROW_FN RFp2 f_sqr_lazy(const RFp2& a) { return f_mul(a, a); }
// a0 b0 + a1 b1 with one reduction per coefficient (a's components < 8p, b's < KB p)
template <int KB0, int KB1>
ROW_FN RFp2 f_sp2(const RFp2& a0, const RFp2& b0, const RFp2& a1, const RFp2& b1) {
    const RowK K = ROW_K();
    rv32 bv[4][13];
    rfp_spread(bv[0], b0.c0);
    rfp_spread(bv[1], b0.c1);
    rfp_spread(bv[2], b1.c0);
    rfp_spread(bv[3], b1.c1);
    const rv32 re[4] = {a0.c0.v, rfp_neg_lazy<8>(a0.c1, K).v, a1.c0.v, rfp_neg_lazy<8>(a1.c1, K).v};
    const rv32 c0 = row_sumprod<4>(re, bv, K.p);
    rv32 bw[4][13];
    for (int i = 0; i < 13; i++) bw[0][i] = bv[1][i], bw[1][i] = bv[0][i], bw[2][i] = bv[3][i], bw[3][i] = bv[2][i];
    const rv32 im[4] = {a0.c0.v, a0.c1.v, a1.c0.v, a1.c1.v};
    return RFp2{RFp{c0}, RFp{row_sumprod<4>(im, bw, K.p)}};
}
ROW_FN bool f_is_zero(const RFp2& a) {
    const RowK K = ROW_K();
    return !rv_test(rv_row_any(rv_or(rfp_canon(a.c0, K).v, rfp_canon(a.c1, K).v)));
}
ROW_FN bool f_eq(const RFp2& a, const RFp2& b) {
    const RowK K = ROW_K();
    return !rv_test(rv_row_any(rv_or(rv_xor(rfp_canon(a.c0, K).v, rfp_canon(b.c0, K).v), rv_xor(rfp_canon(a.c1, K).v, rfp_canon(b.c1, K).v))));
}
ROW_FN void f_set_zero(RFp2& a) { a = RFp2{rfp_zero(), rfp_zero()}; }
ROW_FN void f_set_one(RFp2& a) { a = RFp2{rfp_const(blsc::ONE), rfp_zero()}; }
ROW_FN RFp2 rfp2_conj(const RFp2& a) { const RowK K = ROW_K(); return RFp2{a.c0, rfp_neg(a.c1, K)}; }

}  // namespace ecg
