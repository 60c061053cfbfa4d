// G1 (E1/Fp: y^2 = x^3 + 4) and G2 (E2/Fp2: y^2 = x^3 + 4(1+i)) group arithmetic, ZCash point
// (de)compression and subgroup checks for the gfx950 BLS path.
//
// Reference behaviour being replaced (all inside blst, reached from
// /root/reference/ethereum-consensus/src/crypto/bls.rs):
//   :279-285  PublicKey::try_from  -> blst key_validate   = g1_decompress + reject inf + subgroup
//   :330-336  Signature::try_from  -> blst from_bytes     = g2_decompress (on-curve only)
//   :86-90 / :141-145 aggregate / eth_aggregate_public_keys = decompress + sum + compress
// Formulas are the standard a = 0 Jacobian ones (dbl-2009-l, add-2007-bl, madd-2007-bl); the
// subgroup checks are the endomorphism tests (Scott, ePrint 2021/1130):
//   G1:  phi(P) + P == [x^2] P      (phi(x,y) = (beta x, y) acts as [x^2 - 1] on G1)
//   G2:  psi(Q) == [x] Q
// which cost two / one 64-bit scalar multiplications instead of a 255-bit one.
#pragma once
#include "ecgpu_status.h"
#include "bls_tower.h"

namespace ecg {

// ---- field-generic helpers (overloads over Fp / Fp2) ------------------------------------------
ECG_HD Fp f_add(const Fp& a, const Fp& b) { return fp_add(a, b); }
ECG_HD Fp f_sub(const Fp& a, const Fp& b) { return fp_sub(a, b); }
ECG_HD Fp f_dbl(const Fp& a) { return fp_dbl(a); }
ECG_HD Fp f_neg(const Fp& a) { return fp_neg(a); }
ECG_HD Fp f_mul(const Fp& a, const Fp& b) { return fp_mul(a, b); }
ECG_HD Fp f_sqr(const Fp& a) { return fp_sqr(a); }
ECG_HD bool f_is_zero(const Fp& a) { return fp_is_zero(a); }
ECG_HD bool f_eq(const Fp& a, const Fp& b) { return fp_eq(a, b); }
ECG_HD void f_set_zero(Fp& a) { a = fp_zero(); }
ECG_HD void f_set_one(Fp& a) { a = fp_one(); }
ECG_HD Fp f_inv(const Fp& a) { return fp_inv(a); }

ECG_HD Fp2 f_add(const Fp2& a, const Fp2& b) { return fp2_add(a, b); }
ECG_HD Fp2 f_sub(const Fp2& a, const Fp2& b) { return fp2_sub(a, b); }
ECG_HD Fp2 f_dbl(const Fp2& a) { return fp2_dbl(a); }
ECG_HD Fp2 f_neg(const Fp2& a) { return fp2_neg(a); }
ECG_HD Fp2 f_mul(const Fp2& a, const Fp2& b) { return fp2_mulx(a, b); }
ECG_HD Fp2 f_sqr(const Fp2& a) { return fp2_sqrx(a); }
ECG_HD bool f_is_zero(const Fp2& a) { return fp2_is_zero(a); }
ECG_HD bool f_eq(const Fp2& a, const Fp2& b) { return fp2_eq(a, b); }
ECG_HD void f_set_zero(Fp2& a) { a = fp2_zero(); }
ECG_HD void f_set_one(Fp2& a) { a = fp2_one(); }
ECG_HD Fp2 f_inv(const Fp2& a) { return fp2_inv(a); }

// lazy product operands and sums of two products, generic over the two fields (bounds: bls_fp.h).  f_sp2<KB0, KB1> is
// a0 b0 + a1 b1 with ONE reduction per coefficient; KB0 / KB1 bound the components of b0 / b1 in units of p (Fp2 only:
// the imaginary parts are negated lazily).
ECG_HD Fp f_add_lazy(const Fp& a, const Fp& b) { return fp_add_lazy(a, b); }
ECG_HD Fp2 f_add_lazy(const Fp2& a, const Fp2& b) { return fp2_add_lazy(a, b); }
template <int K>
ECG_HD Fp f_sub_lazy(const Fp& a, const Fp& b) { return fp_sub_lazy_k<K>(a, b); }
template <int K>
ECG_HD Fp2 f_sub_lazy(const Fp2& a, const Fp2& b) { return Fp2{fp_sub_lazy_k<K>(a.c0, b.c0), fp_sub_lazy_k<K>(a.c1, b.c1)}; }
template <int K>
ECG_HD Fp f_neg_lazy(const Fp& a) { return fp_neg_lazy<K>(a); }
template <int K>
ECG_HD Fp2 f_neg_lazy(const Fp2& a) { return Fp2{fp_neg_lazy<K>(a.c0), fp_neg_lazy<K>(a.c1)}; }
// the square of a lazy value < K p; a - 2b in [0, 2p)
template <int K>
ECG_HD Fp f_sqr_lazy(const Fp& a) { return fp_sqr(a); }
template <int K>
ECG_HD Fp2 f_sqr_lazy(const Fp2& a) { return fp2_sqr_lazy<K>(a); }
ECG_HD Fp f_sub_dbl(const Fp& a, const Fp& b) { return fp_sub_dbl(a, b); }
ECG_HD Fp2 f_sub_dbl(const Fp2& a, const Fp2& b) { return fp2_sub_dbl(a, b); }
template <int KB0, int KB1>
ECG_HD Fp f_sp2(const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1) { return fp_sumprod2(a0, b0, a1, b1); }
template <int KB0, int KB1>
ECG_HD Fp2 f_sp2(const Fp2& a0, const Fp2& b0, const Fp2& a1, const Fp2& b1) {
    const Fp2 x[2] = {a0, a1}, y[2] = {b0, b1};
    const Fp ny[2] = {fp_neg_lazy<KB0>(b0.c1), fp_neg_lazy<KB1>(b1.c1)};
    return fp2_sumprod<2>(x, y, ny);
}

template <class F>
struct Jac {
    F x, y, z;
};
template <class F>
struct Aff {
    F x, y;
    u32 inf;
};
typedef Jac<Fp> J1;
typedef Jac<Fp2> J2;
typedef Aff<Fp> A1;
typedef Aff<Fp2> A2;

template <class F>
ECG_HD void jac_set_inf(Jac<F>& r) {
    f_set_one(r.x);
    f_set_one(r.y);
    f_set_zero(r.z);
}
template <class F>
ECG_HD bool jac_is_inf(const Jac<F>& p) {
    return f_is_zero(p.z);
}
template <class F>
ECG_HD void jac_from_aff(Jac<F>& r, const Aff<F>& a) {
    if (a.inf) {
        jac_set_inf(r);
        return;
    }
    r.x = a.x;
    r.y = a.y;
    f_set_one(r.z);
}
template <class F>
ECG_HD void jac_neg(Jac<F>& r, const Jac<F>& p) {
    r.x = p.x;
    r.y = f_neg(p.y);
    r.z = p.z;
}

#if defined(ECG_TOWER_CALLS)
// COMPACT-CODE variant: dbl-2009-l as published (5 squarings, 2 products, modular additions).  inf -> inf; y == 0 -> inf.
template <class F>
ECG_HD void jac_dbl_inl(Jac<F>& r, const Jac<F>& p) {
    F A = f_sqr(p.x);
    F B = f_sqr(p.y);
    F C = f_sqr(B);
    F D = f_sub(f_sub(f_sqr(f_add(p.x, B)), A), C);
    D = f_dbl(D);
    F E = f_add(f_dbl(A), A);
    F Fq = f_sqr(E);
    F Z3 = f_dbl(f_mul(p.y, p.z));
    F X3 = f_sub(Fq, f_dbl(D));
    F C8 = f_dbl(f_dbl(f_dbl(C)));
    r.y = f_sub(f_mul(E, f_sub(D, X3)), C8);
    r.x = X3;
    r.z = Z3;
}
#else
// Doubling (a = 0), the dbl-2009-l quantities regrouped so that no modular addition touches a product:
//   A = X^2, B = Y^2, D = 4 X B, E = 3A (lazy),
//   X3 = E^2 - 2D (one squaring of the lazy E, one correction into [0, 2p)),
//   Y3 = E (D - X3) - 8 B^2 = E (D - X3 + 2p) + (8p - 4B)(2B),   Z3 = (2Y) Z
// 3 squarings, 2 products, 1 sum of two products over lazy operands (the textbook form: 5 squarings, 2 products and 14 modular
// additions / doublings).  Round 3: X3 was the sum of two products E E + (8p - 4X)(2B) -- 1 716 multiplies and 2 380
// instructions in Fp2 where the squaring and the correction take 702 and 1 310.  Bounds (units of p^2, per coefficient; Fp2
// doubles them): E E 36; E (D - X3) 24 + 32 = 56.  inf -> inf (Z3 = 0); y == 0 -> inf.  r may alias p.
template <class F>
ECG_HD void jac_dbl_inl(Jac<F>& r, const Jac<F>& p) {
    const F A = f_sqr(p.x);
    const F B = f_sqr(p.y);
    const F X2 = f_add_lazy(p.x, p.x), X4 = f_add_lazy(X2, X2);  // < 8p
    const F D = f_mul(X4, B);
    const F E = f_add_lazy(f_add_lazy(A, A), A);                 // < 6p
    const F B2 = f_add_lazy(B, B), B4 = f_add_lazy(B2, B2);      // < 4p, < 8p
    const F n4B = f_neg_lazy<8>(B4);                             // 8p - 4B
    const F X3 = f_sub_dbl(f_sqr_lazy<6>(E), D);
    const F Y3 = f_sp2<4, 4>(E, f_sub_lazy<2>(D, X3), n4B, B2);
    const F Z3 = f_mul(f_add_lazy(p.y, p.y), p.z);
    r.x = X3;
    r.y = Y3;
    r.z = Z3;
}
#endif
// the out-of-line forms: every point they are handed is a local of the caller (private segment, see ecg_priv_load)
template <class F>
ECG_HD_NOINLINE void jac_dbl(Jac<F>& r, const Jac<F>& p) {
    const Jac<F> x = ecg_priv_load(p);
    Jac<F> z;
    jac_dbl_inl(z, x);
    ecg_priv_store(r, z);
}

// madd-2007-bl: Jacobian + affine (affine not infinity).  r may alias p.
template <class F>
ECG_HD void jac_add_aff_inl(Jac<F>& r, const Jac<F>& p, const F& qx, const F& qy) {
    if (jac_is_inf(p)) {
        r.x = qx;
        r.y = qy;
        f_set_one(r.z);
        return;
    }
    F Z1Z1 = f_sqr(p.z);
    F U2 = f_mul(qx, Z1Z1);
    F S2 = f_mul(f_mul(qy, p.z), Z1Z1);
    F H = f_sub(U2, p.x);
    F rr = f_sub(S2, p.y);
    if (f_is_zero(H)) {
        if (f_is_zero(rr)) {
            // the doubling is out of line and takes its operand by address: a COPY made here, in the branch, keeps `p`
            // itself a value (its address never escapes, so it lives in registers -- round 4: with `jac_dbl(r, p)` the
            // whole operand was written back to the private segment at entry and re-read piecemeal, 60 k of the 181 k
            // cycles of a G2 addition, profiles/r04f_h2c_parts.txt)
            const Jac<F> pc = p;
            Jac<F> d;
            jac_dbl(d, pc);
            r = d;
        } else {
            jac_set_inf(r);
        }
        return;
    }
    rr = f_dbl(rr);
    F HH = f_sqr(H);
    F I = f_dbl(f_dbl(HH));
    F J = f_mul(H, I);
    F V = f_mul(p.x, I);
    F X3 = f_sub(f_sub(f_sqr(rr), J), f_dbl(V));
    F Y3 = f_sub(f_mul(rr, f_sub(V, X3)), f_dbl(f_mul(p.y, J)));
    F Z3 = f_sub(f_sub(f_sqr(f_add(p.z, H)), Z1Z1), HH);
    r.x = X3;
    r.y = Y3;
    r.z = Z3;
}
template <class F>
ECG_HD_NOINLINE void jac_add_aff(Jac<F>& r, const Jac<F>& p, const F& qx, const F& qy) {
    const Jac<F> x = ecg_priv_load(p);
    const F ax = ecg_priv_load(qx), ay = ecg_priv_load(qy);
    Jac<F> z;
    jac_add_aff_inl(z, x, ax, ay);
    ecg_priv_store(r, z);
}

// add-2007-bl: Jacobian + Jacobian, all special cases.  r may alias p or q.
template <class F>
ECG_HD void jac_add_inl(Jac<F>& r, const Jac<F>& p, const Jac<F>& q) {
    if (jac_is_inf(p)) {
        r = q;
        return;
    }
    if (jac_is_inf(q)) {
        r = p;
        return;
    }
    F Z1Z1 = f_sqr(p.z);
    F Z2Z2 = f_sqr(q.z);
    F U1 = f_mul(p.x, Z2Z2);
    F U2 = f_mul(q.x, Z1Z1);
    F S1 = f_mul(f_mul(p.y, q.z), Z2Z2);
    F S2 = f_mul(f_mul(q.y, p.z), Z1Z1);
    F H = f_sub(U2, U1);
    F rr = f_sub(S2, S1);
    if (f_is_zero(H)) {
        if (f_is_zero(rr)) {
            const Jac<F> pc = p;  // see jac_add_aff_inl
            Jac<F> d;
            jac_dbl(d, pc);
            r = d;
        } else {
            jac_set_inf(r);
        }
        return;
    }
    rr = f_dbl(rr);
    F I = f_sqr(f_dbl(H));
    F J = f_mul(H, I);
    F V = f_mul(U1, I);
    F X3 = f_sub(f_sub(f_sqr(rr), J), f_dbl(V));
    F Y3 = f_sub(f_mul(rr, f_sub(V, X3)), f_dbl(f_mul(S1, J)));
    F Z3 = f_mul(f_sub(f_sub(f_sqr(f_add(p.z, q.z)), Z1Z1), Z2Z2), H);
    r.x = X3;
    r.y = Y3;
    r.z = Z3;
}
template <class F>
ECG_HD_NOINLINE void jac_add(Jac<F>& r, const Jac<F>& p, const Jac<F>& q) {
    const Jac<F> x = ecg_priv_load(p), y = ecg_priv_load(q);
    Jac<F> z;
    jac_add_inl(z, x, y);
    ecg_priv_store(r, z);
}

template <class F>
ECG_HD bool jac_eq(const Jac<F>& p, const Jac<F>& q) {
    bool pi = jac_is_inf(p), qi = jac_is_inf(q);
    if (pi || qi) return pi && qi;
    F Z1Z1 = f_sqr(p.z), Z2Z2 = f_sqr(q.z);
    if (!f_eq(f_mul(p.x, Z2Z2), f_mul(q.x, Z1Z1))) return false;
    return f_eq(f_mul(f_mul(p.y, q.z), Z2Z2), f_mul(f_mul(q.y, p.z), Z1Z1));
}

template <class F>
ECG_HD void jac_to_aff(Aff<F>& r, const Jac<F>& p) {
    if (jac_is_inf(p)) {
        f_set_zero(r.x);
        f_set_zero(r.y);
        r.inf = 1;
        return;
    }
    F zi = f_inv(p.z);
    F zi2 = f_sqr(zi);
    r.x = f_mul(p.x, zi2);
    r.y = f_mul(f_mul(p.y, zi2), zi);
    r.inf = 0;
}

// [|x|] P, |x| = 0xd201000000010000 (MSB-first double-and-add: 63 doublings, 5 additions)
template <class F>
ECG_HD_NOINLINE void jac_mul_xabs(Jac<F>& r, const Jac<F>& p) {
#if defined(ECG_TOWER_CALLS)
    const Jac<F> base = ecg_priv_load(p);
    Jac<F> acc = base;
    for (int b = 62; b >= 0; b--) {
        jac_dbl(acc, acc);
        if ((blsc::X_ABS >> b) & 1) jac_add(acc, acc, base);
    }
    ecg_priv_store(r, acc);
#else
    // The running point stays in registers across the 63 doublings (inlined: an out-of-line doubling reloads its operand with
    // 21 separate waits on the private segment, a fifth of its time on a lone wave); the base point is needed five times and
    // lives in memory -- 78 more live dwords under the doubling would come back as spills (cf. fp12_cyc_pow_x).
    Jac<F> base_mem;
    Jac<F> acc = ecg_priv_load(p);
    ecg_priv_store(base_mem, acc);
    for (int b = 62; b >= 0; b--) {
        jac_dbl_inl(acc, acc);
        if ((blsc::X_ABS >> b) & 1) {
            Jac<F> t = acc;
            jac_add(t, t, base_mem);
            acc = t;
        }
    }
    ecg_priv_store(r, acc);
#endif
}

// the same for an AFFINE base (the subgroup checks start from a decoded point): the five additions are mixed ones
// (7 products + 4 squarings instead of 11 + 5)
template <class F>
ECG_HD_NOINLINE void jac_mul_xabs_aff(Jac<F>& r, const Aff<F>& p) {
#if defined(ECG_TOWER_CALLS)
    Jac<F> b;
    jac_from_aff(b, p);
    jac_mul_xabs(r, b);
#else
    F bx_mem, by_mem;
    const Aff<F> a = ecg_priv_load(p);
    ecg_priv_store(bx_mem, a.x);
    ecg_priv_store(by_mem, a.y);
    Jac<F> acc;
    jac_from_aff(acc, a);
    for (int b = 62; b >= 0; b--) {
        jac_dbl_inl(acc, acc);
        if ((blsc::X_ABS >> b) & 1) {
            Jac<F> t = acc;
            jac_add_aff(t, t, bx_mem, by_mem);
            acc = t;
        }
    }
    ecg_priv_store(r, acc);
#endif
}

// [k] P for a scalar of `nwords` 32-bit LE words (test-vector generation: sk -> pk, signing)
template <class F>
ECG_HD_NOINLINE void jac_mul_scalar(Jac<F>& r, const Jac<F>& p, const u32* k, int nwords) {
    const Jac<F> base = ecg_priv_load(p);
    Jac<F> acc;
    jac_set_inf(acc);
    for (int b = nwords * 32 - 1; b >= 0; b--) {
        jac_dbl(acc, acc);
        if ((k[b >> 5] >> (b & 31)) & 1) jac_add(acc, acc, base);
    }
    ecg_priv_store(r, acc);
}

// ---- endomorphisms and subgroup checks --------------------------------------------------------
ECG_HD bool g1_in_subgroup(const A1& p) {
    if (p.inf) return true;
    J1 P, t, lhs;
    jac_from_aff(P, p);
    jac_mul_xabs_aff(t, p);
    jac_mul_xabs(t, t);  // [x^2] P
    Fp bx = fp_mul(p.x, blsc::BETA);
    jac_add_aff(lhs, P, bx, p.y);  // P + phi(P)
    return jac_eq(lhs, t);
}

ECG_HD void g2_psi(J2& r, const J2& p) {
    r.x = fp2_mulx(fp2_conj(p.x), blsc::PSI_X);
    r.y = fp2_mulx(fp2_conj(p.y), blsc::PSI_Y);
    r.z = fp2_conj(p.z);
}

ECG_HD bool g2_in_subgroup(const A2& q) {
    if (q.inf) return true;
    J2 Q, t, ps;
    jac_from_aff(Q, q);
    jac_mul_xabs_aff(t, q);
    jac_neg(t, t);  // [x] Q, x < 0
    g2_psi(ps, Q);
    return jac_eq(ps, t);
}

// ---- ZCash compressed encodings -> BLST_ERROR codes ---------------------
// 0 SUCCESS, 1 BAD_ENCODING, 2 POINT_NOT_ON_CURVE, 3 POINT_NOT_IN_GROUP (x == 0), as blst's
// Uncompress does; infinity decodes to inf = 1 with SUCCESS (the callers decide what it means).
ECG_HD bool bytes_all_zero(const u8* b, int from, int to) {
    u32 o = 0;
    for (int i = from; i < to; i++) o |= b[i];
    return o == 0;
}

ECG_HD int g1_decompress_inl(A1& r, const u8* b) {
    r.inf = 0;
    r.x = fp_zero();
    r.y = fp_zero();
    const u32 b0 = b[0];
    if (!(b0 & 0x80)) return ECGPU_BAD_ENCODING;
    if (b0 & 0x40) {
        if ((b0 & 0x3f) == 0 && bytes_all_zero(b, 1, 48)) {
            r.inf = 1;
            return ECGPU_SUCCESS;
        }
        return ECGPU_BAD_ENCODING;
    }
    Fp raw = raw_from_be48(b, true);
    if (raw_geq(raw, blsc::P)) return ECGPU_BAD_ENCODING;
    Fp x = fp_from_raw(raw);
    Fp y;
    if (!fp_sqrt(fp_add(fp_mul(fp_sqr(x), x), blsc::B1), y)) return ECGPU_POINT_NOT_ON_CURVE;
    if (fp_lex_largest(y) != ((b0 & 0x20) != 0)) y = fp_neg(y);
    if (fp_is_zero(x)) return ECGPU_POINT_NOT_IN_GROUP;
    r.x = x;
    r.y = y;
    return ECGPU_SUCCESS;
}
ECG_HD_NOINLINE int g1_decompress(A1& r, const u8* b) {  // r: a local of the caller (private segment); b: global memory
    A1 z;
    const int rc = g1_decompress_inl(z, b);
    ecg_priv_store(r, z);
    return rc;
}

ECG_HD void g1_compress(u8* out, const A1& p) {
    if (p.inf) {
        out[0] = 0xc0;
        for (int i = 1; i < 48; i++) out[i] = 0;
        return;
    }
    raw_to_be48(fp_to_raw(p.x), out);
    out[0] |= 0x80;
    if (fp_lex_largest(p.y)) out[0] |= 0x20;
}

ECG_HD int g2_decompress_inl(A2& r, const u8* b) {
    r.inf = 0;
    r.x = fp2_zero();
    r.y = fp2_zero();
    const u32 b0 = b[0];
    if (!(b0 & 0x80)) return ECGPU_BAD_ENCODING;
    if (b0 & 0x40) {
        if ((b0 & 0x3f) == 0 && bytes_all_zero(b, 1, 96)) {
            r.inf = 1;
            return ECGPU_SUCCESS;
        }
        return ECGPU_BAD_ENCODING;
    }
    Fp r1 = raw_from_be48(b, true);
    Fp r0 = raw_from_be48(b + 48, false);
    if (raw_geq(r1, blsc::P) || raw_geq(r0, blsc::P)) return ECGPU_BAD_ENCODING;
    Fp2 x = Fp2{fp_from_raw(r0), fp_from_raw(r1)};
    Fp2 y;
    if (!fp2_sqrt(fp2_add(fp2_mulx(fp2_sqrx(x), x), blsc::B2), y)) return ECGPU_POINT_NOT_ON_CURVE;
    if (fp2_lex_largest(y) != ((b0 & 0x20) != 0)) y = fp2_neg(y);
    if (fp2_is_zero(x)) return ECGPU_POINT_NOT_IN_GROUP;
    r.x = x;
    r.y = y;
    return ECGPU_SUCCESS;
}
ECG_HD_NOINLINE int g2_decompress(A2& r, const u8* b) {  // r: a local of the caller (private segment); b: global memory
    A2 z;
    const int rc = g2_decompress_inl(z, b);
    ecg_priv_store(r, z);
    return rc;
}

ECG_HD void g2_compress(u8* out, const A2& p) {
    if (p.inf) {
        out[0] = 0xc0;
        for (int i = 1; i < 96; i++) out[i] = 0;
        return;
    }
    raw_to_be48(fp_to_raw(p.x.c1), out);
    raw_to_be48(fp_to_raw(p.x.c0), out + 48);
    out[0] |= 0x80;
    if (fp2_lex_largest(p.y)) out[0] |= 0x20;
}

// blst key_validate (crypto/bls.rs:279-285): decode, reject infinity, subgroup check
ECG_HD int g1_key_validate(A1& r, const u8* b) {
    int st = g1_decompress(r, b);
    if (st) return st;
    if (r.inf) return ECGPU_PK_IS_INFINITY;
    if (!g1_in_subgroup(r)) return ECGPU_POINT_NOT_IN_GROUP;
    return ECGPU_SUCCESS;
}

}  // namespace ecg
