// G2 point arithmetic with one point per 16-lane row (bls_rowfield.h supplies the field interface; the Jacobian routines are
// the generic ones of bls_curve.h, every special case included): the end of the message stage -- the two mapped points added,
// the cofactor cleared (Budroni-Pintore: two multiplications by |x|, 126 doublings and 15 additions), the affine conversion --
// as a kernel for small batches.  hash_to_curve is what blst performs inside every verify call
// (/root/reference/ethereum-consensus/src/crypto/bls.rs:71,126).
#pragma once
#include "bls_rowpair.h"
#include "bls_h2c.h"

namespace ecg {

typedef Jac<RFp2> RJ2;
typedef Jac<RP2> PJ2;

// psi(x, y, z) = (conj(x) PSI_X, conj(y) PSI_Y, conj(z)); F = RFp2 (a point per row) or RP2 (a point per row pair)
template <class F>
ROW_FN void r_g2_psi(Jac<F>& r, const Jac<F>& p) {
    r.x = f_mul(f_conj(p.x), f_const2((const F*)nullptr, blsc::PSI_X));
    r.y = f_mul(f_conj(p.y), f_const2((const F*)nullptr, blsc::PSI_Y));
    r.z = f_conj(p.z);
}
// [x^2 - x - 1] P + [x - 1] psi(P) + psi^2(2P), term by term as g2_clear_cofactor (bls_h2c.h)
template <class F>
ECG_HD_NOINLINE void r_g2_clear_cofactor(Jac<F>& r, const Jac<F>& p_in) {
    const Jac<F> p = p_in;
    Jac<F> t1, t2, t3, n;
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
// ---- a G2 doubling over BOTH row pairs of a wave (round 5, last): one message per wave ---------------------------------------
// A doubling on a row pair is six products one after the other (jac_dbl_inl of bls_curve.h: A = X^2, B = Y^2, D = 4 X B, E^2,
// Y3 as a sum of two products, Z3 = 2 Y Z), but its dependency depth is three.  With the point held in BOTH pairs of the wave
// (lanes 0 .. 31 and 32 .. 63, the same values) every step runs ONE instruction sequence on operands selected per pair:
//     step 1   pair 0: A = X^2                          pair 1: B = Y^2
//     step 2   pair 0: E^2, E = 3A                      pair 1: D = (4X) B
//     step 3   pair 0: Y3 = E (D - X3) + (8p - 4B)(2B)  pair 1: Z3 = (2Y) Z + 0 * 0
// with four lane exchanges between the pairs (D and B over to pair 0; X3, Y3 | Z3 back to both): 8 product-iterations instead of
// 14, ~1 100 instructions instead of ~1 700 -- for the 126 doublings of the cofactor clearing of a LONE message, whose second row
// pair would otherwise sit idle.  Bounds as in jac_dbl_inl (the unused halves compute in-range garbage: every operand of a
// product is < 8p in both pairs).  inf -> inf (Z3 = 0); y == 0 -> inf.  r may alias p.
ROW_FN RP2 rq_sel(const RP2& pair0, const RP2& pair1) { return RP2{rv_sel(rv_quad_hi(), pair1.v, pair0.v)}; }
ROW_FN RP2 rq_other(const RP2& a) { return RP2{rv_other_pair(a.v)}; }
ROW_FN void jac_dbl_quad(PJ2& r, const PJ2& p) {
    const RP2 AB = f_sqr(rq_sel(p.x, p.y));                            // A | B
    const RP2 T3 = f_add_lazy(f_add_lazy(AB, AB), AB);                  // E = 3A < 6p | (3B)
    const RP2 X2 = f_add_lazy(p.x, p.x), X4 = f_add_lazy(X2, X2);       // 4X < 8p
    const RP2 ED = f_mul(rq_sel(T3, X4), rq_sel(T3, AB));               // E^2 | D
    const RP2 D0 = rq_other(ED), B0 = rq_other(AB);                     // D | (E^2),  B | (A)
    const RP2 X3 = f_sub_dbl(ED, D0);                                   // E^2 - 2D in [0, 2p) | (in-range garbage)
    const RP2 B2 = f_add_lazy(B0, B0), B4 = f_add_lazy(B2, B2);         // < 4p, < 8p
    const RP2 n4B = f_neg_lazy<8>(B4);
    RP2 zero;
    f_set_zero(zero);
    const RP2 YZ = f_sp2<4, 4>(rq_sel(T3, f_add_lazy(p.y, p.y)), rq_sel(f_sub_lazy<2>(D0, X3), p.z), rq_sel(n4B, zero), rq_sel(B2, zero));  // Y3 | Z3
    const RP2 ZY = rq_other(YZ), X3o = rq_other(X3);
    r.x = rq_sel(X3, X3o);
    r.y = rq_sel(YZ, ZY);
    r.z = rq_sel(ZY, YZ);
}
// An addition over both pairs (add-2007-bl with every special case, as jac_add_inl of bls_curve.h): 16 products in 8 steps
//     Z1Z1 | Z2Z2;   U2 = X2 Z1Z1 | U1 = X1 Z2Z2;   Y2 Z1 | Y1 Z2;   S2 | S1;   I = (2H)^2 | (Z1 + Z2)^2;
//     J = H I | Z3 = (..) H;   V = U1 I | r^2;   r (V - X3) | S1 J
// with seven exchanges; H and r = 2 (S2 - S1) are formed in both pairs, so the special cases branch the same way in both.
// p, q and the result are held in both pairs.  r may alias p or q.
ROW_FN void jac_add_quad(PJ2& r, const PJ2& p, const PJ2& q) {
    if (jac_is_inf(p)) {
        r = q;
        return;
    }
    if (jac_is_inf(q)) {
        r = p;
        return;
    }
    const RP2 zsel = rq_sel(p.z, q.z);
    const RP2 ZZ = f_sqr(zsel);                                   // Z1Z1 | Z2Z2
    const RP2 U = f_mul(rq_sel(q.x, p.x), ZZ);                    // U2 | U1
    const RP2 S = f_mul(f_mul(rq_sel(q.y, p.y), zsel), ZZ);       // S2 | S1
    const RP2 Uo = rq_other(U), So = rq_other(S), ZZo = rq_other(ZZ);
    const RP2 U1 = rq_sel(Uo, U), S1 = rq_sel(So, S);             // (in both pairs from here)
    const RP2 H = f_sub(rq_sel(U, Uo), U1);
    RP2 rr = f_sub(rq_sel(S, So), S1);
    if (f_is_zero(H)) {
        if (f_is_zero(rr)) {
            const PJ2 pc = p;
            jac_dbl_quad(r, pc);
        } else {
            jac_set_inf(r);
        }
        return;
    }
    rr = f_dbl(rr);
    const RP2 IW = f_sqr(rq_sel(f_dbl(H), f_add(p.z, q.z)));      // I | (Z1 + Z2)^2
    const RP2 IWo = rq_other(IW);
    const RP2 I = rq_sel(IW, IWo), W = rq_sel(IWo, IW);
    const RP2 zt = f_sub(f_sub(W, rq_sel(ZZ, ZZo)), rq_sel(ZZo, ZZ));
    const RP2 JZ = f_mul(rq_sel(H, zt), rq_sel(I, H));            // J | Z3
    const RP2 VR = f_mul(rq_sel(U1, rr), rq_sel(I, rr));          // V | r^2
    const RP2 JZo = rq_other(JZ), VRo = rq_other(VR);
    const RP2 J = rq_sel(JZ, JZo), V = rq_sel(VR, VRo);
    const RP2 X3 = f_sub(f_sub(rq_sel(VRo, VR), J), f_dbl(V));
    const RP2 YY = f_mul(rq_sel(rr, S1), rq_sel(f_sub(V, X3), J));  // r (V - X3) | S1 J
    const RP2 YYo = rq_other(YY);
    r.x = X3;
    r.y = f_sub(rq_sel(YY, YYo), f_dbl(rq_sel(YYo, YY)));
    r.z = rq_sel(JZo, JZ);
}
// [|x|] P with the doublings and the additions over both pairs
ECG_HD_NOINLINE void jac_mul_xabs_quad(PJ2& r, const PJ2& p_in) {
    const PJ2 base = p_in;
    PJ2 acc = base;
    for (int b = 62; b >= 0; b--) {
        jac_dbl_quad(acc, acc);
        if ((blsc::X_ABS >> b) & 1) {
            PJ2 t = acc;
            jac_add_quad(t, t, base);
            acc = t;
        }
    }
    r = acc;
}
// r_g2_clear_cofactor with its two multiplications by |x| on both pairs
ECG_HD_NOINLINE void r_g2_clear_cofactor_quad(PJ2& r, const PJ2& p_in) {
    const PJ2 p = p_in;
    PJ2 t1, t2, t3, n;
    jac_mul_xabs_quad(t1, p);
    jac_neg(t1, t1);  // [x] P
    r_g2_psi(t2, p);  // psi(P)
    jac_dbl_quad(t3, p);
    r_g2_psi(t3, t3);
    r_g2_psi(t3, t3);  // psi^2(2P)
    jac_neg(n, t2);
    jac_add_quad(t3, t3, n);   // psi^2(2P) - psi(P)
    jac_add_quad(t2, t1, t2);  // [x] P + psi(P)
    jac_mul_xabs_quad(t2, t2);
    jac_neg(t2, t2);  // [x^2] P + [x] psi(P)
    jac_add_quad(t3, t3, t2);
    jac_neg(n, t1);
    jac_add_quad(t3, t3, n);
    jac_neg(n, p);
    jac_add_quad(t3, t3, n);
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
ROW_FN RFp2 f_inv_tab(const RFp2& a, u32* tab) { return rfp2_inv(a, tab); }
ROW_FN RFp2 f_load2(const RFp2*, const Fp2* src) { return rfp2_load(src); }
ROW_FN void f_store2(Fp2* dst, const RFp2& a) {
    rfp_store(&dst->c0, a.c0);
    rfp_store(&dst->c1, a.c1);
}
ROW_FN bool f_first_lane(const RFp2*) {
#if defined(__HIPCC__)
    return (threadIdx.x & 15u) == 0;
#else
    return true;
#endif
}

// a value every lane of the row holds in full (computed by the one-lane routines, the same in all lanes) -> limb per lane
ROW_FN RFp rfp_of(const Fp& x) {
#if defined(__HIPCC__)
    const u32 l = threadIdx.x & 15u;
    u32 v = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) v = l == (u32)i ? x.l[i] : v;
    return RFp{v};
#else
    return RFp{row_const_limb(x.l)};
#endif
}
ROW_FN RFp2 rfp2_of(const Fp2& x) { return RFp2{rfp_of(x.c0), rfp_of(x.c1)}; }

// ---- square roots, signs ------------------------------------------------------------------------------------------------------
// s with s^2 == a (true) for a square; s = a^((p+1)/4) either way (fp_sqrt)
ROW_FN bool rfp_sqrt(const RFp& a, RFp& s, u32* tab, const RowK& K) {
    s = rfp_mul(rfp_pow_pm3d4(a, tab, K), a, K);
    return rfp_eq(rfp_sqr(s, K), a, K);
}
// fp2_sqrt_with_norm_root (bls_fp.h): s^2 = norm(a), a1 != 0; ONE exponentiation
ROW_FN bool rfp2_sqrt_with_norm_root(const RFp2& a, const RFp& s, RFp2& r, u32* tab, const RowK& K) {
    const RFp inv2 = rfp_const(blsc::INV2);
    const RFp d = rfp_mul(rfp_add(a.c0, s, K), inv2, K);
    const RFp t = rfp_pow_pm3d4(d, tab, K);
    const RFp c = rfp_mul(t, d, K);
    const RFp ha1t = rfp_mul(rfp_mul(a.c1, inv2, K), t, K);
    const bool first = rfp_eq(rfp_sqr(c, K), d, K);
    r.c0 = first ? c : rfp_neg(ha1t, K);
    r.c1 = first ? ha1t : c;
    return f_eq(f_sqr(r), a);
}
// fp2_sqrt (bls_fp.h): any root; true iff a is a square
ROW_FN bool rfp2_sqrt(const RFp2& a, RFp2& r, u32* tab, const RowK& K) {
    if (rfp_is_zero(a.c1, K)) {
        RFp s;
        if (rfp_sqrt(a.c0, s, tab, K)) {
            r = RFp2{s, rfp_zero()};
            return true;
        }
        const bool ok = rfp_sqrt(rfp_neg(a.c0, K), s, tab, K);
        r = RFp2{rfp_zero(), s};
        return ok;
    }
    rv32 bv[2][13];
    rfp_spread(bv[0], a.c0);
    rfp_spread(bv[1], a.c1);
    const rv32 sq[2] = {a.c0.v, a.c1.v};
    const RFp n{row_sumprod<2>(sq, bv, K.p)};
    RFp s;
    if (!rfp_sqrt(n, s, tab, K)) return false;
    return rfp2_sqrt_with_norm_root(a, s, r, tab, K);
}
// the canonical integer (out of Montgomery form): limbs exact, in [0, p)
ROW_FN RFp rfp_to_raw(const RFp& a, const RowK& K) {
    const RFp one_raw{rv_eq(K.lane, rv_splat(0))};  // the integer 1: limb 0 = 1
    return rfp_canon(rfp_mul(a, one_raw, K), K);
}
// RFC 9380 sgn0 (m = 2)
ROW_FN u32 rfp2_sgn0(const RFp2& a, const RowK& K) {
    const RFp r0 = rfp_to_raw(a.c0, K), r1 = rfp_to_raw(a.c1, K);
    const bool z0 = !rv_test(rv_row_any(r0.v));
    const u32 b0 = rv_test(rv_and(rv_bcast<0>(r0.v), rv_splat(1))) ? 1u : 0u, b1 = rv_test(rv_and(rv_bcast<0>(r1.v), rv_splat(1))) ? 1u : 0u;
    return b0 | ((z0 ? 1u : 0u) & b1);
}
// raw > (p - 1) / 2 (the ZCash sign of an Fp)
ROW_FN bool rfp_lex_largest(const RFp& a, const RowK& K) {
    const RFp raw = rfp_to_raw(a, K);
    const rv32 half = row_const_limb(blsc::HALF_P);
    // half - raw < 0  <=>  raw > half: signed limbs, exact carries, sign of the top limb
    rv64 t = rv_mad64s(rv_splat(1), half, rv_zero64());
    t = rv_mad64s(rv_splat((u32)-1), raw.v, t);
    return rv_test(rv_shr(rv_bcast<12>(r_norm_signed(t, K)), 31));
}
ROW_FN bool rfp2_lex_largest(const RFp2& a, const RowK& K) {
    if (!rfp_is_zero(a.c1, K)) return rfp_lex_largest(a.c1, K);
    return rfp_lex_largest(a.c0, K);
}

// ---- the SSWU map and the 3-isogeny on a row (map_to_curve_g2 of bls_h2c.h, step by step) -------------------------------------
ROW_FN RFp2 rfp2_horner(const Fp2* c, int deg, const RFp2& x) {
    RFp2 acc = rfp2_const(c[deg]);
    for (int i = deg - 1; i >= 0; i--) acc = f_add(f_mul(acc, x), rfp2_const(c[i]));
    return acc;
}
ROW_FN RFp rfp_norm(const RFp2& a, const RowK& K) {
    rv32 bv[2][13];
    rfp_spread(bv[0], a.c0);
    rfp_spread(bv[1], a.c1);
    const rv32 sq[2] = {a.c0.v, a.c1.v};
    return RFp{row_sumprod<2>(sq, bv, K.p)};
}
// u, tv2(u) and sgn0(u) come from the one-lane prologue (every lane of the row computed them in full).
// (round 6, last) NO inversion of tv2: x1 = -B (tv2 + 1) / (A tv2) = xn / xd stays a fraction until the exponentiation the map
// needs anyway has run.  g(x1) = gn / xd^3 with gn = xn^3 + A xn xd^2 + B xd^3; with D = norm(xd), V = norm(gn) D (the
// squareness of norm(g(x1)) = norm(gn) / D^3, D^4 being a square) and T = V^((p-3)/4):
//     V T   = V^((p+1)/4)            -> (V T)^2 == V  <=>  g(x1) is a square in Fp2
//     V T^2 = V^((p-1)/2) = chi      -> 1 / V = chi T^2,  1 / D = norm(gn) / V
//     norm(g(x1))^((p+1)/4) = V T / D^2                                     (D^(p+1) = D^2)
// so ONE exponentiation yields the root of the norm the Fp2 square root starts from AND 1 / xd = conj(xd) / D: the 80 us
// one-lane division steps of 1 / tv2 (0.42 ms of the map before, 0.34 after) are gone.  V == 0 (g(x1) == 0: no hash output
// in practice) takes the old road through the one-lane inversion.
ROW_FN void r_map_to_curve_g2(RJ2& r, const RFp2& u, const RFp2& tv2, const Fp2& tv2_lane, bool tv2_zero, u32 sgn_u, u32* tab) {
    const RowK K = row_k();
    const RFp2 tv1 = f_mul(rfp2_const(blsc::SSWU_Z), f_sqr(u));
    const RFp2 A = rfp2_const(blsc::SSWU_A), B = rfp2_const(blsc::SSWU_B);
    RFp2 x1, one;
    f_set_one(one);
    RFp sn;
    bool sq1 = false, have_sn = false;
    if (tv2_zero) {
        x1 = rfp2_const(blsc::SSWU_B_OVER_ZA);
    } else {
        const RFp2 xn = f_neg(f_mul(B, f_add(tv2, one))), xd = f_mul(A, tv2);
        const RFp2 xd2 = f_sqr(xd);
        const RFp2 gn = f_add(f_add(f_mul(f_sqr(xn), xn), f_mul(A, f_mul(xn, xd2))), f_mul(B, f_mul(xd2, xd)));
        const RFp D = rfp_norm(xd, K), Ngn = rfp_norm(gn, K);
        const RFp V = rfp_mul(Ngn, D, K);
        if (!rfp_is_zero(V, K)) {
            const RFp T = rfp_pow_pm3d4(V, tab, K);
            const RFp VT = rfp_mul(V, T, K);
            sq1 = rfp_eq(rfp_sqr(VT, K), V, K);
            const RFp T2 = rfp_sqr(T, K);
            const RFp invD = rfp_mul(Ngn, sq1 ? T2 : rfp_neg(T2, K), K);
            const RFp2 t = f_mul(xn, RFp2{xd.c0, rfp_neg(xd.c1, K)});
            x1 = RFp2{rfp_mul(t.c0, invD, K), rfp_mul(t.c1, invD, K)};
            sn = rfp_mul(VT, rfp_sqr(invD, K), K);
            have_sn = true;
        } else {
            x1 = f_mul(rfp2_const(blsc::SSWU_MB_OVER_A), f_add(one, rfp2_of(fp2_inv(tv2_lane))));
        }
    }
    const RFp2 gx1 = f_add(f_add(f_mul(f_sqr(x1), x1), f_mul(A, x1)), B);
    const RFp2 x2 = f_mul(tv1, x1);
    const RFp2 gx2 = f_add(f_add(f_mul(f_sqr(x2), x2), f_mul(A, x2)), B);
    if (!have_sn) sq1 = rfp_sqrt(rfp_norm(gx1, K), sn, tab, K);
    if (!sq1) {
        const RFp m = rfp_norm(tv1, K);
        const RFp v = rfp_mul(rfp_const(blsc::SQRT_M5), rfp_norm(u, K), K);
        sn = rfp_mul(rfp_mul(m, sn, K), v, K);
    }
    RFp2 x = sq1 ? x1 : x2, y;
    const RFp2 g = sq1 ? gx1 : gx2;
    bool ok = !rfp_is_zero(g.c1, K) && rfp2_sqrt_with_norm_root(g, sn, y, tab, K);
    if (!ok) {
        x = x1;
        if (!rfp2_sqrt(gx1, y, tab, K)) {
            x = x2;
            (void)rfp2_sqrt(gx2, y, tab, K);
        }
    }
    if (sgn_u != rfp2_sgn0(y, K)) y = f_neg(y);
    const RFp2 xn = rfp2_horner(blsc::ISO_XNUM, 3, x), xd = rfp2_horner(blsc::ISO_XDEN, 2, x);
    const RFp2 yn = rfp2_horner(blsc::ISO_YNUM, 3, x), yd = rfp2_horner(blsc::ISO_YDEN, 3, x);
    if (f_is_zero(xd) || f_is_zero(yd)) {
        jac_set_inf(r);
        return;
    }
    const RFp2 yd2 = f_sqr(yd);
    r.x = f_mul(f_mul(xn, xd), yd2);
    r.y = f_mul(f_mul(f_mul(y, yn), f_mul(f_sqr(xd), xd)), yd2);
    r.z = f_mul(xd, yd);
}
// one map of one message: the one-lane prologue (expand_message_xmd, the field element, tv2: every lane of the row computes
// them in full, which costs a lone wave nothing) and the map on the row; the point goes to memory with exact limbs
ROW_FN void r_hash_to_g2_map(J2* out, const u8* msg, size_t msg_len, int j, u32* tab) {
    Fp2 u0, u1;
    hash_to_field2(u0, u1, msg, msg_len);
    const Fp2 u = j ? u1 : u0;
    const Fp2 t = sswu_tv2(u);
    const bool tz = fp2_is_zero(t);
    RJ2 q;
    r_map_to_curve_g2(q, rfp2_of(u), rfp2_of(t), t, tz, fp2_sgn0(u), tab);
    rfp_store(&out->x.c0, q.x.c0);
    rfp_store(&out->x.c1, q.x.c1);
    rfp_store(&out->y.c0, q.y.c0);
    rfp_store(&out->y.c1, q.y.c1);
    rfp_store(&out->z.c0, q.z.c0);
    rfp_store(&out->z.c1, q.z.c1);
}

// the psi subgroup check of a decoded signature (g2_in_subgroup of bls_curve.h): psi(Q) == [x] Q
template <class F>
ROW_FN bool r_g2_in_subgroup(const A2* q) {
    if (q->inf) return true;
    Aff<F> a{f_load2((const F*)nullptr, &q->x), f_load2((const F*)nullptr, &q->y), 0u};
    Jac<F> Q, t, ps;
    jac_from_aff(Q, a);
    jac_mul_xabs_aff(t, a);
    jac_neg(t, t);
    r_g2_psi(ps, Q);
    return jac_eq(ps, t);
}

// psi(Q) == [x] Q for an affine point already in row-pair registers
ROW_FN bool r_g2_in_subgroup_regs(const Aff<RP2>& a) {
    Jac<RP2> Q, t, ps;
    jac_from_aff(Q, a);
    jac_mul_xabs_aff(t, a);
    jac_neg(t, t);
    r_g2_psi(ps, Q);
    return jac_eq(ps, t);
}

// Signature::try_from + the group check of verify for ONE signature on a row pair (crypto/bls.rs:330-336, 71; g2_decompress_inl
// and g2_in_subgroup of bls_curve.h, step by step): the flag bytes, the range check and the conversion of x to Montgomery
// form by the one-lane routines in every lane (a lone wave pays nothing for the redundancy), x^3 + 4(1 + i) and its square root
// -- two Fp exponentiations -- on the row (each row of the pair holds both components and computes the same thing), the ZCash
// sign, then the psi check with one component per row.  *sd: the decoding's status; *sg: ECGPU_POINT_NOT_IN_GROUP or 0; the
// affine point goes to memory exactly as the one-lane decoder leaves it (zero coordinates unless the decoding succeeded).
ROW_FN void r_sig_decode_and_group(A2* out, u8* sd, u8* sg, const u8* b, u32* tab) {
    const RowK K = row_k();
    const RP2* tag = nullptr;
    const bool first = f_first_lane(tag);
    RFp2 x, y;
    f_set_zero(x);
    f_set_zero(y);
    u32 inf = 0;
    int st = ECGPU_SUCCESS;
    const u32 b0 = b[0];
    if (!(b0 & 0x80)) {
        st = ECGPU_BAD_ENCODING;
    } else if (b0 & 0x40) {
        if ((b0 & 0x3f) == 0 && bytes_all_zero(b, 1, 96)) inf = 1;
        else st = ECGPU_BAD_ENCODING;
    } else {
        const Fp r1 = raw_from_be48(b, true), r0 = raw_from_be48(b + 48, false);
        if (raw_geq(r1, blsc::P) || raw_geq(r0, blsc::P)) {
            st = ECGPU_BAD_ENCODING;
        } else {
            const RFp2 xr = rfp2_of(Fp2{fp_from_raw(r0), fp_from_raw(r1)});
            const RFp2 g = f_add(f_mul(f_sqr(xr), xr), rfp2_const(blsc::B2));
            RFp2 yr;
            f_set_zero(yr);
            if (!rfp2_sqrt(g, yr, tab, K)) {
                st = ECGPU_POINT_NOT_ON_CURVE;
            } else {
                if (rfp2_lex_largest(yr, K) != ((b0 & 0x20) != 0)) yr = f_neg(yr);
                if (f_is_zero(xr)) st = ECGPU_POINT_NOT_IN_GROUP;
                else x = xr, y = yr;
            }
        }
    }
    // every lane of the pair holds both components: row 0 of the pair stores (the row-pair store of f_store2 takes one component per row)
    const RP2 xp{rv_sel(rv_pair_row(), x.c1.v, x.c0.v)}, yp{rv_sel(rv_pair_row(), y.c1.v, y.c0.v)};
    f_store2(&out->x, xp);
    f_store2(&out->y, yp);
    u8 g8 = 0;
    if (st == ECGPU_SUCCESS && !inf && !r_g2_in_subgroup_regs(Aff<RP2>{xp, yp, 0u})) g8 = ECGPU_POINT_NOT_IN_GROUP;
    if (first) {
        out->inf = inf;
        *sd = (u8)st;
        *sg = g8;
    }
}

// PublicKey::try_from = blst key_validate for ONE key on a row (crypto/bls.rs:279-285; g1_decompress_inl + the infinity rule +
// g1_in_subgroup): the square root -- one Fp exponentiation -- and the endomorphism check on the row
ROW_FN void r_pk_validate(A1* out, u8* st_out, const u8* b, u32* tab) {
    const RowK K = row_k();
    RFp x = rfp_zero(), y = rfp_zero();
    u32 inf = 0;
    int st = ECGPU_SUCCESS;
    const u32 b0 = b[0];
    if (!(b0 & 0x80)) {
        st = ECGPU_BAD_ENCODING;
    } else if (b0 & 0x40) {
        if ((b0 & 0x3f) == 0 && bytes_all_zero(b, 1, 48)) inf = 1, st = ECGPU_PK_IS_INFINITY;
        else st = ECGPU_BAD_ENCODING;
    } else {
        const Fp raw = raw_from_be48(b, true);
        if (raw_geq(raw, blsc::P)) {
            st = ECGPU_BAD_ENCODING;
        } else {
            const RFp xr = rfp_of(fp_from_raw(raw));
            const RFp g = rfp_add(rfp_mul(rfp_sqr(xr, K), xr, K), rfp_const(blsc::B1), K);
            RFp yr;
            if (!rfp_sqrt(g, yr, tab, K)) {
                st = ECGPU_POINT_NOT_ON_CURVE;
            } else {
                if (rfp_lex_largest(yr, K) != ((b0 & 0x20) != 0)) yr = rfp_neg(yr, K);
                if (rfp_is_zero(xr, K)) st = ECGPU_POINT_NOT_IN_GROUP;
                else x = xr, y = yr;
            }
        }
    }
    rfp_store(&out->x, x);
    rfp_store(&out->y, y);
    if (st == ECGPU_SUCCESS) {
        const Aff<R1> a{R1{x.v}, R1{y.v}, 0u};
        Jac<R1> P, t, lhs;
        jac_from_aff(P, a);
        jac_mul_xabs_aff(t, a);
        jac_mul_xabs(t, t);  // [x^2] P
        const R1 bx = f_mul(a.x, R1{row_const_limb(blsc::BETA.l)});
        jac_add_aff(lhs, P, bx, a.y);  // P + phi(P)
        if (!jac_eq(lhs, t)) st = ECGPU_POINT_NOT_IN_GROUP;
    }
#if defined(__HIPCC__)
    if ((threadIdx.x & 15u) == 0) {
        out->inf = inf;
        *st_out = (u8)st;
    }
#else
    out->inf = inf;
    *st_out = (u8)st;
#endif
}

// the subgroup check of a decoded public key on a row (g1_in_subgroup of bls_curve.h): P + phi(P) == [x^2] P
ROW_FN bool r_g1_in_subgroup(const A1* p) {
    if (p->inf) return true;
    const Aff<R1> a{R1{rfp_load(&p->x).v}, R1{rfp_load(&p->y).v}, 0u};
    Jac<R1> P, t, lhs;
    jac_from_aff(P, a);
    jac_mul_xabs_aff(t, a);
    jac_mul_xabs(t, t);  // [x^2] P
    const R1 bx = f_mul(a.x, R1{row_const_limb(blsc::BETA.l)});
    jac_add_aff(lhs, P, bx, a.y);  // P + phi(P)
    return jac_eq(lhs, t);
}

// q0 + q1, cofactor, affine: H(m) to memory (pointers uniform over the row / the row pair)
template <class F>
ROW_FN void r_hash_to_g2_finish(A2* out, const J2* q0, const J2* q1, u32* tab) {
    const F* tag = nullptr;
    Jac<F> a{f_load2(tag, &q0->x), f_load2(tag, &q0->y), f_load2(tag, &q0->z)};
    const Jac<F> b{f_load2(tag, &q1->x), f_load2(tag, &q1->y), f_load2(tag, &q1->z)};
    jac_add(a, a, b);
    r_g2_clear_cofactor(a, a);
    const bool inf = jac_is_inf(a);
    F x, y;
    f_set_zero(x);
    f_set_zero(y);
    if (!inf) {
        const F zi = f_inv_tab(a.z, tab);
        const F zi2 = f_sqr(zi);
        x = f_mul(a.x, zi2);
        y = f_mul(f_mul(a.y, zi2), zi);
    }
    f_store2(&out->x, x);
    f_store2(&out->y, y);
    if (f_first_lane(tag)) out->inf = inf ? 1u : 0u;
}

// ... the same with one message per WAVE: both row pairs hold the point, the doublings run over both (jac_dbl_quad)
ROW_FN void r_hash_to_g2_finish_quad(A2* out, const J2* q0, const J2* q1, u32* tab) {
    const RP2* tag = nullptr;
    PJ2 a{f_load2(tag, &q0->x), f_load2(tag, &q0->y), f_load2(tag, &q0->z)};
    const PJ2 b{f_load2(tag, &q1->x), f_load2(tag, &q1->y), f_load2(tag, &q1->z)};
    jac_add_quad(a, a, b);
    r_g2_clear_cofactor_quad(a, a);
    const bool inf = jac_is_inf(a);
    RP2 x, y;
    f_set_zero(x);
    f_set_zero(y);
    if (!inf) {
        const RP2 zi = f_inv_tab(a.z, tab);
        const RP2 zi2 = f_sqr(zi);
        x = f_mul(a.x, zi2);
        y = f_mul(f_mul(a.y, zi2), zi);
    }
    f_store2(&out->x, x);  // (both pairs store the same limbs)
    f_store2(&out->y, y);
    if (f_first_lane(tag)) out->inf = inf ? 1u : 0u;
}

}  // namespace ecg
