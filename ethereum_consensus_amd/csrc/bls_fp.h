// Fp and Fp2 arithmetic for BLS12-381 on gfx950 (replaces blst's fp/fp2 layer that sits under
// /root/reference/ethereum-consensus/src/crypto/bls.rs:4 `blst::min_pk`).
//
// One lane = one field operation.  Fp = 13 x 30-bit limbs, Montgomery R = 2^390, values kept
// "almost reduced" (< 2p, limbs < 2^30): see bls_types.h for why.  The Montgomery product is
// operand scanning over 64-bit column accumulators: 13 rows x (13 a*b + 13 m*p) independent
// v_mad_u64_u32 + 13 v_mul_lo_u32 for the quotient digits = 351 quarter-rate multiplies and
// ~200 full-rate ops per Fp product -- the unit the BLS roofline in DESIGN.md is priced in.
// fp_mul / fp_sqr are real calls (ECG_HD_NOINLINE): one body per kernel instead of one per use,
// which keeps the pairing kernels inside the instruction cache and the build in seconds.
#pragma once
#include "bls_consts.h"

namespace ecg {

constexpr u32 FP_MASK = 0x3fffffffu;
constexpr int FP_N = 13;

// op census for the roofline arithmetic in DESIGN.md / bench.py: host lane simulator only
#if !defined(__HIPCC__) && defined(ECG_COUNT_OPS)
extern unsigned long long g_ecg_fp_mul_count, g_ecg_fp_sqr_count;
#define ECG_COUNT_MUL() (++g_ecg_fp_mul_count)
#define ECG_COUNT_SQR() (++g_ecg_fp_sqr_count)
#else
#define ECG_COUNT_MUL() ((void)0)
#define ECG_COUNT_SQR() ((void)0)
#endif

// ---------------------------------------------------------------------------------------------
// Fp
// ---------------------------------------------------------------------------------------------
ECG_HD Fp fp_zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = 0;
    return r;
}
ECG_HD Fp fp_one() { return blsc::ONE; }

// r = a - k if a >= k else a   (k = p or 2p, normalized limbs)
ECG_HD Fp fp_cond_sub(const Fp& a, const u32* k) {
    Fp d;
    int32_t bw = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        int32_t t = (int32_t)a.l[i] - (int32_t)k[i] + bw;
        d.l[i] = (u32)t & FP_MASK;
        bw = t >> 30;
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = bw < 0 ? a.l[i] : d.l[i];
    return r;
}
// the unique representative in [0, p)
ECG_HD Fp fp_canon(const Fp& a) { return fp_cond_sub(a, blsc::P); }

ECG_HD bool fp_is_zero(const Fp& a) {
    u32 z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        z |= a.l[i];
        e |= a.l[i] ^ blsc::P[i];
    }
    return z == 0 || e == 0;
}
ECG_HD bool fp_eq(const Fp& a, const Fp& b) {
    Fp x = fp_canon(a), y = fp_canon(b);
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) o |= x.l[i] ^ y.l[i];
    return o == 0;
}

ECG_HD Fp fp_add(const Fp& a, const Fp& b) {
    Fp s;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        u32 t = a.l[i] + b.l[i] + c;
        s.l[i] = t & FP_MASK;
        c = t >> 30;
    }
    return fp_cond_sub(s, blsc::P2);  // a + b < 4p -> < 2p
}

ECG_HD Fp fp_sub(const Fp& a, const Fp& b) {
    Fp d;
    int32_t bw = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        int32_t t = (int32_t)a.l[i] - (int32_t)b.l[i] + bw;
        d.l[i] = (u32)t & FP_MASK;
        bw = t >> 30;
    }
    const u32 m = (u32)bw;  // all-ones when a < b: add 2p back (a - b > -2p)
    u32 c = 0;
    Fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        u32 t = d.l[i] + (blsc::P2[i] & m) + c;
        r.l[i] = t & FP_MASK;
        c = t >> 30;
    }
    return r;
}

// Lazy forms for PRODUCT OPERANDS ONLY: limbs renormalized (< 2^30), no modular correction.  The Montgomery
// product needs a * b < R p = 632 p^2 (R = 2^390) for a result < 2p, so operands may grow to 16p x 16p; every
// use below states the bound it relies on.  39 / 52 instructions instead of 104.
ECG_HD Fp fp_add_lazy(const Fp& a, const Fp& b) {  // a + b
    Fp s;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        u32 t = a.l[i] + b.l[i] + c;
        s.l[i] = i + 1 < FP_N ? (t & FP_MASK) : t;
        c = t >> 30;
    }
    return s;
}
ECG_HD Fp fp_sub_lazy(const Fp& a, const Fp& b) {  // a - b + 2p in (0, a + 2p) for b < 2p
    Fp s;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        int32_t t = (int32_t)a.l[i] - (int32_t)b.l[i] + (int32_t)blsc::P2[i] + c;
        s.l[i] = i + 1 < FP_N ? ((u32)t & FP_MASK) : (u32)t;
        c = t >> 30;
    }
    return s;
}

ECG_HD Fp fp_neg(const Fp& a) { return fp_sub(fp_zero(), a); }
ECG_HD Fp fp_dbl(const Fp& a) { return fp_add(a, a); }

// Montgomery product a*b/R mod p (result < 2p for a, b < 2p; raw inputs up to 2^384 are fine too).
ECG_HD Fp fp_mul_body(const Fp& a, const Fp& b) {
    u64 T[27];
#pragma unroll
    for (int i = 0; i < 27; i++) T[i] = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        const u32 bi = b.l[i];
#pragma unroll
        for (int j = 0; j < FP_N; j++) T[i + j] += (u64)a.l[j] * bi;
        const u32 m = ((u32)T[i] * blsc::N0) & FP_MASK;
#pragma unroll
        for (int j = 0; j < FP_N; j++) T[i + j] += (u64)m * blsc::P[j];
        T[i + 1] += T[i] >> 30;  // T[i] == 0 mod 2^30 now
        if (i == 6) {
            // 14 products per column so far; renormalize so that rows 7..12 (12 more) still fit
#pragma unroll
            for (int c = 7; c <= 18; c++) {
                T[c + 1] += T[c] >> 30;
                T[c] &= FP_MASK;
            }
        }
    }
    Fp r;
#pragma unroll
    for (int c = 13; c < 25; c++) {
        T[c + 1] += T[c] >> 30;
        r.l[c - 13] = (u32)T[c] & FP_MASK;
    }
    r.l[12] = (u32)T[25];
    return r;
}
#if defined(__HIP_DEVICE_COMPILE__)
// The out-of-line call takes its operands as two 13-element vectors: clang's AMDGPU ABI gives a
// function 16 argument registers for aggregates, so the second `Fp` struct of fp_mul(Fp, Fp) travelled
// through the stack (13 dwords of scratch store + load per product); vectors are passed in VGPRs.
typedef u32 fp_vec13 __attribute__((ext_vector_type(13)));
static __device__ __attribute__((noinline)) fp_vec13 fp_mul_call(fp_vec13 a, fp_vec13 b) {
    Fp x, y;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        x.l[i] = a[i];
        y.l[i] = b[i];
    }
    const Fp r = fp_mul_body(x, y);
    fp_vec13 o;
#pragma unroll
    for (int i = 0; i < FP_N; i++) o[i] = r.l[i];
    return o;
}
ECG_HD Fp fp_mul(const Fp& a, const Fp& b) {
    fp_vec13 x, y;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        x[i] = a.l[i];
        y[i] = b.l[i];
    }
    const fp_vec13 o = fp_mul_call(x, y);
    Fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = o[i];
    return r;
}
#else
ECG_HD_NOINLINE Fp fp_mul(Fp a, Fp b) {
    ECG_COUNT_MUL();
    return fp_mul_body(a, b);
}
#endif

// Montgomery square: 91 a_i*a_j products (off-diagonal ones doubled) + one carry pass, then the
// 13 reduction rows: 260 multiplies instead of 351.
ECG_HD_NOINLINE Fp fp_sqr(Fp a) {
    ECG_COUNT_SQR();
    u64 T[27];
#pragma unroll
    for (int i = 0; i < 27; i++) T[i] = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        T[2 * i] += (u64)a.l[i] * a.l[i];
        const u32 a2 = a.l[i] << 1;  // < 2^31: at most 7 products of < 2^61 per column
#pragma unroll
        for (int j = i + 1; j < FP_N; j++) T[i + j] += (u64)a2 * a.l[j];
    }
#pragma unroll
    for (int c = 0; c < 25; c++) {
        T[c + 1] += T[c] >> 30;
        T[c] &= FP_MASK;
    }
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        const u32 m = ((u32)T[i] * blsc::N0) & FP_MASK;
#pragma unroll
        for (int j = 0; j < FP_N; j++) T[i + j] += (u64)m * blsc::P[j];
        T[i + 1] += T[i] >> 30;
    }
    Fp r;
#pragma unroll
    for (int c = 13; c < 25; c++) {
        T[c + 1] += T[c] >> 30;
        r.l[c - 13] = (u32)T[c] & FP_MASK;
    }
    r.l[12] = (u32)T[25];
    return r;
}

// small-constant multiples
ECG_HD Fp fp_mul3(const Fp& a) { return fp_add(fp_dbl(a), a); }

// a^e for a public 384-bit exponent (12 LE words), 4-bit fixed window.
ECG_HD_NOINLINE Fp fp_pow(Fp a, const u32* e) {
    Fp tab[16];
    tab[0] = fp_one();
    tab[1] = a;
    for (int i = 2; i < 16; i++) tab[i] = fp_mul(tab[i - 1], a);
    Fp r = fp_one();
    bool started = false;
    for (int w = 95; w >= 0; w--) {
        u32 nib = (e[w >> 3] >> ((w & 7) * 4)) & 15;
        if (started) {
            r = fp_sqr(r);
            r = fp_sqr(r);
            r = fp_sqr(r);
            r = fp_sqr(r);
        }
        if (nib) {
            r = started ? fp_mul(r, tab[nib]) : tab[nib];
            started = true;
        }
    }
    return r;
}

ECG_HD Fp fp_inv(const Fp& a) { return fp_pow(a, blsc::EXP_INV); }  // 0 -> 0

// Square root for p = 3 mod 4.  Returns true and s with s^2 == a when a is a square.
// Also hands back t = a^((p-3)/4): when a is a non-zero square, t == 1/s.
ECG_HD bool fp_sqrt_inv(const Fp& a, Fp& s, Fp& inv_s) {
    inv_s = fp_pow(a, blsc::EXP_PM3D4);
    s = fp_mul(inv_s, a);
    return fp_eq(fp_sqr(s), a);
}
ECG_HD bool fp_sqrt(const Fp& a, Fp& s) {
    Fp t;
    return fp_sqrt_inv(a, s, t);
}

// Montgomery <-> plain ("raw" = the integer itself in 13 x 30-bit limbs)
ECG_HD Fp fp_from_raw(const Fp& raw) { return fp_mul(raw, blsc::R2); }  // raw < 2^384 is enough
ECG_HD Fp fp_to_raw(const Fp& a) {  // canonical integer in [0, p)
    Fp one = fp_zero();
    one.l[0] = 1;
    return fp_canon(fp_mul(a, one));
}

// raw comparisons against normalized constants
ECG_HD bool raw_geq(const Fp& a, const u32* k) {  // a >= k
    int32_t bw = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) bw = ((int32_t)a.l[i] - (int32_t)k[i] + bw) >> 30;
    return bw >= 0;
}
ECG_HD bool raw_gt(const Fp& a, const u32* k) {  // a > k
    int32_t bw = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) bw = ((int32_t)k[i] - (int32_t)a.l[i] + bw) >> 30;
    return bw < 0;
}

// 12 little-endian 32-bit words <-> 13 x 30-bit limbs
ECG_HD Fp raw_from_words(const u32* w) {
    Fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        const int bit = 30 * i, q = bit >> 5, sh = bit & 31;
        u32 v = w[q] >> sh;
        if (sh > 2 && q + 1 < 12) v |= w[q + 1] << (32 - sh);
        r.l[i] = v & FP_MASK;
    }
    return r;
}
ECG_HD void raw_to_words(const Fp& r, u32* w) {
#pragma unroll
    for (int q = 0; q < 12; q++) {
        const int bit = 32 * q, i = bit / 30, sh = bit % 30;  // word q starts inside limb i
        u32 v = r.l[i] >> sh;
        if (i + 1 < FP_N) v |= r.l[i + 1] << (30 - sh);
        if (sh > 28 && i + 2 < FP_N) v |= r.l[i + 2] << (60 - sh);
        w[q] = v;
    }
}
// 48 big-endian bytes -> raw limbs (no reduction, no Montgomery).  `mask` clears the three ZCash
// flag bits of byte 0.
ECG_HD Fp raw_from_be48(const u8* b, bool mask) {
    u32 w[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const u8* q = b + 4 * (11 - i);
        u32 b0 = q[0];
        if (mask && i == 11) b0 &= 0x1f;
        w[i] = (b0 << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
    return raw_from_words(w);
}
ECG_HD void raw_to_be48(const Fp& r, u8* b) {
    u32 w[12];
    raw_to_words(r, w);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        u8* q = b + 4 * (11 - i);
        q[0] = (u8)(w[i] >> 24);
        q[1] = (u8)(w[i] >> 16);
        q[2] = (u8)(w[i] >> 8);
        q[3] = (u8)w[i];
    }
}

// ZCash sign bit of an Fp: canonical value > (p-1)/2
ECG_HD bool fp_lex_largest(const Fp& a) { return raw_gt(fp_to_raw(a), blsc::HALF_P); }

// ---------------------------------------------------------------------------------------------
// Fp2 = Fp[i]/(i^2 + 1)
// ---------------------------------------------------------------------------------------------
ECG_HD Fp2 fp2_zero() { return Fp2{fp_zero(), fp_zero()}; }
ECG_HD Fp2 fp2_one() { return Fp2{fp_one(), fp_zero()}; }
ECG_HD bool fp2_is_zero(const Fp2& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
ECG_HD bool fp2_eq(const Fp2& a, const Fp2& b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
ECG_HD Fp2 fp2_add(const Fp2& a, const Fp2& b) { return Fp2{fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; }
ECG_HD Fp2 fp2_sub(const Fp2& a, const Fp2& b) { return Fp2{fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; }
ECG_HD Fp2 fp2_neg(const Fp2& a) { return Fp2{fp_neg(a.c0), fp_neg(a.c1)}; }
ECG_HD Fp2 fp2_dbl(const Fp2& a) { return Fp2{fp_dbl(a.c0), fp_dbl(a.c1)}; }
ECG_HD Fp2 fp2_conj(const Fp2& a) { return Fp2{a.c0, fp_neg(a.c1)}; }
ECG_HD Fp2 fp2_mul3(const Fp2& a) { return fp2_add(fp2_dbl(a), a); }
// (a0 + a1 i)(1 + i) = (a0 - a1) + (a0 + a1) i
ECG_HD Fp2 fp2_mul_xi(const Fp2& a) { return Fp2{fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1)}; }
ECG_HD Fp2 fp2_mul_fp(const Fp2& a, const Fp& k) { return Fp2{fp_mul(a.c0, k), fp_mul(a.c1, k)}; }
// a + b as a product operand (components < bound(a) + bound(b), see fp_add_lazy)
ECG_HD Fp2 fp2_add_lazy(const Fp2& a, const Fp2& b) { return Fp2{fp_add_lazy(a.c0, b.c0), fp_add_lazy(a.c1, b.c1)}; }

// Karatsuba: 3 Fp products.  Operand components may be lazy sums < 8p: the inner sums are then < 16p and
// 16p x 16p = 256 p^2 < 632 p^2; every product comes back < 2p.
ECG_HD Fp2 fp2_mul(const Fp2& a, const Fp2& b) {
    Fp t0 = fp_mul(a.c0, b.c0);
    Fp t1 = fp_mul(a.c1, b.c1);
    Fp t2 = fp_mul(fp_add_lazy(a.c0, a.c1), fp_add_lazy(b.c0, b.c1));
    return Fp2{fp_sub(t0, t1), fp_sub(fp_sub(t2, t0), t1)};
}
// (a0 + a1)(a0 - a1) + 2 a0 a1 i: 2 Fp products.  Operand components < 2p (a0 - a1 + 2p < 4p, a0 + a1 < 4p).
ECG_HD Fp2 fp2_sqr(const Fp2& a) {
    Fp t0 = fp_mul(fp_add_lazy(a.c0, a.c1), fp_sub_lazy(a.c0, a.c1));
    Fp t1 = fp_mul(a.c0, a.c1);
    return Fp2{t0, fp_dbl(t1)};
}
ECG_HD Fp2 fp2_inv(const Fp2& a) {
    Fp d = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
    return Fp2{fp_mul(a.c0, d), fp_neg(fp_mul(a.c1, d))};
}

// Square root in Fp2 by the norm ("complex") method, given s with s^2 = norm(a) = a0^2 + a1^2 and a1 != 0.
// ONE Fp exponentiation: with d = (a0 + s)/2, t = d^((p-3)/4) and c = t d, either c^2 = d (then x0 = c, 1/x0 = t) or
// c^2 = -d -- and then the OTHER candidate (a0 - s)/2 = a1^2 / (4 c^2) is the square, with root a1 / (2c) = -a1 t / 2
// (c t = d^((p-1)/2) = -1) and imaginary part a1 / (2 x0) = c.  No second exponentiation, no data-dependent branch
// around one: the lanes of a wave stay together.  Any root; false if r^2 != a.
ECG_HD bool fp2_sqrt_with_norm_root(const Fp2& a, const Fp& s, Fp2& r) {
    const Fp d = fp_mul(fp_add(a.c0, s), blsc::INV2);
    const Fp t = fp_pow(d, blsc::EXP_PM3D4);
    const Fp c = fp_mul(t, d);
    const Fp ha1t = fp_mul(fp_mul(a.c1, blsc::INV2), t);  // a1 t / 2
    const bool first = fp_eq(fp_sqr(c), d);
    r.c0 = first ? c : fp_neg(ha1t);
    r.c1 = first ? ha1t : c;
    return fp2_eq(fp2_sqr(r), a);
}

// Square root in Fp2; true iff a is a square.  Any root.  Two Fp exponentiations (norm root, then the above).
ECG_HD_NOINLINE bool fp2_sqrt(Fp2 a, Fp2& r) {
    if (fp_is_zero(a.c1)) {
        Fp s;
        if (fp_sqrt(a.c0, s)) {
            r = Fp2{s, fp_zero()};
            return true;
        }
        // -a0 is then a square (p = 3 mod 4): (s i)^2 = -s^2 = a0
        bool ok = fp_sqrt(fp_neg(a.c0), s);
        r = Fp2{fp_zero(), s};
        return ok;
    }
    Fp n = fp_add(fp_sqr(a.c0), fp_sqr(a.c1));
    Fp s;
    if (!fp_sqrt(n, s)) return false;
    return fp2_sqrt_with_norm_root(a, s, r);
}

// RFC 9380 sgn0 (m = 2) and the ZCash sign of an Fp2 (compare c1 first, then c0)
ECG_HD u32 fp2_sgn0(const Fp2& a) {
    Fp r0 = fp_to_raw(a.c0), r1 = fp_to_raw(a.c1);
    u32 z0 = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) z0 |= r0.l[i];
    z0 = z0 == 0 ? 1u : 0u;
    return (r0.l[0] & 1) | (z0 & (r1.l[0] & 1));
}
ECG_HD bool fp2_lex_largest(const Fp2& a) {
    if (!fp_is_zero(a.c1)) return fp_lex_largest(a.c1);
    return fp_lex_largest(a.c0);
}

}  // namespace ecg
