// Fp and Fp2 arithmetic for BLS12-381 on gfx950 (replaces blst's fp/fp2 layer that sits under
// /root/reference/ethereum-consensus/src/crypto/bls.rs:4 `blst::min_pk`).
//
// One lane = one field operation.  Fp = 13 x 30-bit limbs, Montgomery R = 2^390, values kept
// "almost reduced" (< 2p, limbs < 2^30): see bls_types.h for why.  The Montgomery product is
// operand scanning over 64-bit column accumulators: 13 rows x (13 a*b + 13 m*p) independent
// v_mad_u64_u32 + 13 v_mul_lo_u32 for the quotient digits = 351 multiplies and ~200 cheap ops per Fp product.
// Two forms are built on it: fp_mul / fp_sqr, real calls (one body per kernel: exponentiation chains, G1 arithmetic), and
// fp_sumprod<N>, an inline sum of N products with ONE reduction over lazily reduced operands -- what the Fp2 / Fp6 / Fp12
// tower and the G2 formulas are made of, because on this machine a modular addition costs a sixth of a product
// (DESIGN.md 3.1).
#pragma once
#include "bls_consts.h"

namespace ecg {

constexpr u32 FP_MASK = 0x3fffffffu;
constexpr int FP_N = 13;

// op census for the roofline arithmetic in DESIGN.md / bench.py: host lane simulator only
#if !defined(__HIPCC__) && defined(ECG_COUNT_OPS)
extern unsigned long long g_ecg_fp_mul_count, g_ecg_fp_sqr_count, g_ecg_fp_mad_count;
#define ECG_COUNT_MUL() (++g_ecg_fp_mul_count)
#define ECG_COUNT_SQR() (++g_ecg_fp_sqr_count)
#define ECG_COUNT_MAD(n) (g_ecg_fp_mad_count += (n))  // multiply instructions of the sums of products
#else
#define ECG_COUNT_MUL() ((void)0)
#define ECG_COUNT_SQR() ((void)0)
#define ECG_COUNT_MAD(n) ((void)0)
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

ECG_HD Fp fp_add_inl(const Fp& a, const Fp& b) {
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

ECG_HD Fp fp_sub_inl(const Fp& a, const Fp& b) {
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

#if defined(__HIP_DEVICE_COMPILE__) && defined(ECG_TOWER_CALLS) && defined(ECG_LINEAR_CALLS)
// compact build: the modular additions are calls too (two 13-element vectors in, one out: ~30 instructions per use instead
// of ~100), so that the Miller iteration of the compact kernels fits the 64 KB instruction cache
typedef u32 fp_lin_vec13 __attribute__((ext_vector_type(13)));
static __device__ __attribute__((noinline)) fp_lin_vec13 fp_add_call(fp_lin_vec13 a, fp_lin_vec13 b) {
    Fp x, y;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        x.l[i] = a[i];
        y.l[i] = b[i];
    }
    const Fp r = fp_add_inl(x, y);
    fp_lin_vec13 o;
#pragma unroll
    for (int i = 0; i < FP_N; i++) o[i] = r.l[i];
    return o;
}
static __device__ __attribute__((noinline)) fp_lin_vec13 fp_sub_call(fp_lin_vec13 a, fp_lin_vec13 b) {
    Fp x, y;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        x.l[i] = a[i];
        y.l[i] = b[i];
    }
    const Fp r = fp_sub_inl(x, y);
    fp_lin_vec13 o;
#pragma unroll
    for (int i = 0; i < FP_N; i++) o[i] = r.l[i];
    return o;
}
#define ECG_LIN_CALL(fn, a, b)                                          \
    fp_lin_vec13 x_, y_;                                                \
    _Pragma("unroll") for (int i = 0; i < FP_N; i++) {                  \
        x_[i] = (a).l[i];                                               \
        y_[i] = (b).l[i];                                               \
    }                                                                   \
    const fp_lin_vec13 o_ = fn(x_, y_);                                 \
    Fp r_;                                                              \
    _Pragma("unroll") for (int i = 0; i < FP_N; i++) r_.l[i] = o_[i];   \
    return r_;
ECG_HD Fp fp_add(const Fp& a, const Fp& b) { ECG_LIN_CALL(fp_add_call, a, b) }
ECG_HD Fp fp_sub(const Fp& a, const Fp& b) { ECG_LIN_CALL(fp_sub_call, a, b) }
#else
ECG_HD Fp fp_add(const Fp& a, const Fp& b) { return fp_add_inl(a, b); }
ECG_HD Fp fp_sub(const Fp& a, const Fp& b) { return fp_sub_inl(a, b); }
#endif
ECG_HD Fp fp_neg(const Fp& a) { return fp_sub(fp_zero(), a); }
ECG_HD Fp fp_dbl(const Fp& a) { return fp_add(a, a); }

#if defined(__HIP_DEVICE_COMPILE__)
// nothing but memory and scalar instructions may be scheduled across (operand loads should still be issued early)
#define ECG_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0x3f4)
// T[k] (+)= a[k] * b for the 13 columns of a row, written as the instructions themselves.  Two reasons: from
// `(u64)a * b` LLVM keeps every operand limb that has more than one use as a zero-extended 64-bit register PAIR (the
// extension is CSE'd and then allocated), which doubles the operand registers of a sum of products and spills it to the
// private segment; and one statement per multiply-add gets an s_nop between every two of them (the hazard recogniser's
// conservative rule for back-to-back inline asm), which costs a full issue slot each at one wave per SIMD -- hence ONE
// statement per row (27 operands; round 2 had two, 7 + 6 columns).  The carry-out of v_mad_u64_u32 goes to vcc and is never
// used (columns have 4 bits of headroom).
// FRESH = first column of the row that has not been written yet (13: none).  A column's first multiply-add takes the
// constant 0 as its addend instead of a register pair somebody had to clear: the 27 v_mov_b64 per sum of products that the
// zero-initialised accumulators cost in round 2 (4-5 % of the instructions of a sum of 2 or 3 products) are gone.
#define ECG_MAD_ACC(k, a) "v_mad_u64_u32 %" #k ", vcc, %" #a ", %26, %" #k "\n\t"
#define ECG_MAD_NEW(k, a) "v_mad_u64_u32 %" #k ", vcc, %" #a ", %26, 0\n\t"
#define ECG_ROW_IN(a, b) \
    "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(b)
template <int FRESH>
ECG_D void ecg_mad_row(u64* T, const u32* a, u32 b) {
    static_assert(FRESH == 13 || FRESH == 12 || FRESH == 0, "a row starts a sum (0), opens one new column (12) or none (13)");
    if constexpr (FRESH == 13) {
        asm(ECG_MAD_ACC(0, 13) ECG_MAD_ACC(1, 14) ECG_MAD_ACC(2, 15) ECG_MAD_ACC(3, 16) ECG_MAD_ACC(4, 17) ECG_MAD_ACC(5, 18) ECG_MAD_ACC(6, 19)
                ECG_MAD_ACC(7, 20) ECG_MAD_ACC(8, 21) ECG_MAD_ACC(9, 22) ECG_MAD_ACC(10, 23) ECG_MAD_ACC(11, 24) "v_mad_u64_u32 %12, vcc, %25, %26, %12"
            : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]), "+v"(T[4]), "+v"(T[5]), "+v"(T[6]), "+v"(T[7]), "+v"(T[8]), "+v"(T[9]), "+v"(T[10]),
              "+v"(T[11]), "+v"(T[12])
            : ECG_ROW_IN(a, b)
            : "vcc");
    } else if constexpr (FRESH == 12) {
        asm(ECG_MAD_ACC(0, 13) ECG_MAD_ACC(1, 14) ECG_MAD_ACC(2, 15) ECG_MAD_ACC(3, 16) ECG_MAD_ACC(4, 17) ECG_MAD_ACC(5, 18) ECG_MAD_ACC(6, 19)
                ECG_MAD_ACC(7, 20) ECG_MAD_ACC(8, 21) ECG_MAD_ACC(9, 22) ECG_MAD_ACC(10, 23) ECG_MAD_ACC(11, 24) "v_mad_u64_u32 %12, vcc, %25, %26, 0"
            : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]), "+v"(T[4]), "+v"(T[5]), "+v"(T[6]), "+v"(T[7]), "+v"(T[8]), "+v"(T[9]), "+v"(T[10]),
              "+v"(T[11]), "=&v"(T[12])
            : ECG_ROW_IN(a, b)
            : "vcc");
    } else {
        asm(ECG_MAD_NEW(0, 13) ECG_MAD_NEW(1, 14) ECG_MAD_NEW(2, 15) ECG_MAD_NEW(3, 16) ECG_MAD_NEW(4, 17) ECG_MAD_NEW(5, 18) ECG_MAD_NEW(6, 19)
                ECG_MAD_NEW(7, 20) ECG_MAD_NEW(8, 21) ECG_MAD_NEW(9, 22) ECG_MAD_NEW(10, 23) ECG_MAD_NEW(11, 24) "v_mad_u64_u32 %12, vcc, %25, %26, 0"
            : "=&v"(T[0]), "=&v"(T[1]), "=&v"(T[2]), "=&v"(T[3]), "=&v"(T[4]), "=&v"(T[5]), "=&v"(T[6]), "=&v"(T[7]), "=&v"(T[8]), "=&v"(T[9]),
              "=&v"(T[10]), "=&v"(T[11]), "=&v"(T[12])
            : ECG_ROW_IN(a, b)
            : "vcc");
    }
}
// the same with the limbs of p (compile-time constants, kept in scalar registers: one per instruction is allowed)
ECG_D void ecg_mad_row_p(u64* T, u32 m) {
    asm(ECG_MAD_ACC(0, 13) ECG_MAD_ACC(1, 14) ECG_MAD_ACC(2, 15) ECG_MAD_ACC(3, 16) ECG_MAD_ACC(4, 17) ECG_MAD_ACC(5, 18) ECG_MAD_ACC(6, 19)
            ECG_MAD_ACC(7, 20) ECG_MAD_ACC(8, 21) ECG_MAD_ACC(9, 22) ECG_MAD_ACC(10, 23) ECG_MAD_ACC(11, 24) "v_mad_u64_u32 %12, vcc, %25, %26, %12"
        : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]), "+v"(T[4]), "+v"(T[5]), "+v"(T[6]), "+v"(T[7]), "+v"(T[8]), "+v"(T[9]), "+v"(T[10]),
          "+v"(T[11]), "+v"(T[12])
        : "s"(blsc::P[0]), "s"(blsc::P[1]), "s"(blsc::P[2]), "s"(blsc::P[3]), "s"(blsc::P[4]), "s"(blsc::P[5]), "s"(blsc::P[6]), "s"(blsc::P[7]),
          "s"(blsc::P[8]), "s"(blsc::P[9]), "s"(blsc::P[10]), "s"(blsc::P[11]), "s"(blsc::P[12]), "v"(m)
        : "vcc");
}
// Carry-save step between two columns: next += 4 * (col >> 32), col &= 2^32 - 1 (2^32 = 4 * 2^30).  The high DWORD of the
// column is a register of its own, so moving it is ONE multiply-add (by the constant 4) plus clearing it -- instead of a
// 64-bit shift, a 64-bit addition, a mask and the clear (round 2: 4 instructions per column, a quarter of the non-multiply
// instructions of the tower).  The low dword keeps up to 32 bits instead of 30: 2^32 + 2^34 is still nothing against the
// 2^64 - 15 * 2^60 of headroom.  FRESH: `next` has not been written yet.
template <bool FRESH>
ECG_D void ecg_col_pass_hi(u64& next, u64& col) {
    const u32 hi = (u32)(col >> 32);
    if constexpr (FRESH)
        asm("v_mad_u64_u32 %0, vcc, %1, 4, 0" : "=v"(next) : "v"(hi) : "vcc");
    else
        asm("v_mad_u64_u32 %0, vcc, %1, 4, %0" : "+v"(next) : "v"(hi) : "vcc");
    col = (u64)(u32)col;
}
#else
#define ECG_SCHED_FENCE() ((void)0)
// host lane simulator (CPU test-suite): the same column arithmetic, with every accumulation checked for 64-bit overflow
extern unsigned long long g_ecg_column_overflows;
template <int FRESH>
ECG_HD void ecg_mad_row(u64* T, const u32* a, u32 b) {
    for (int j = 0; j < 13; j++) {
        if (j >= FRESH) T[j] = 0;
        if (__builtin_add_overflow(T[j], (u64)a[j] * b, &T[j])) g_ecg_column_overflows++;
    }
}
ECG_HD void ecg_mad_row_p(u64* T, u32 m) {
    for (int j = 0; j < 13; j++)
        if (__builtin_add_overflow(T[j], (u64)m * blsc::P[j], &T[j])) g_ecg_column_overflows++;
}
template <bool FRESH>
ECG_HD void ecg_col_pass_hi(u64& next, u64& col) {
    if (FRESH) next = 0;
    if (__builtin_add_overflow(next, (col >> 32) << 2, &next)) g_ecg_column_overflows++;
    col &= 0xffffffffull;
}
#endif
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
            // 14 products per column so far; carry-save pass (top down, see ecg_col_pass_hi) so that rows 7..12 (12 more) still fit
#pragma unroll
            for (int c = 18; c >= 7; c--) ecg_col_pass_hi<false>(T[c + 1], T[c]);
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
// k * p in normalized limbs, evaluated at compile time: the offsets of the lazy subtractions below.
struct FpConst {
    u32 l[13];
};
constexpr FpConst fp_p_times(u32 k) {
    FpConst r{};
    u64 c = 0;
    for (int i = 0; i < 13; i++) {
        c += (u64)blsc::P[i] * k;
        r.l[i] = i + 1 < 13 ? (u32)(c & 0x3fffffffu) : (u32)c;
        if (i + 1 < 13) c >>= 30;
    }
    return r;
}
// K p - a in (0, K p] for a < K p: the lazy negation used to fold signs into sums of products.  Limbs renormalized.
template <int K>
ECG_HD Fp fp_neg_lazy(const Fp& a) {
    constexpr FpConst kp = fp_p_times(K);
    Fp s;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        int32_t t = (int32_t)kp.l[i] - (int32_t)a.l[i] + c;
        s.l[i] = i + 1 < FP_N ? ((u32)t & FP_MASK) : (u32)t;
        c = t >> 30;
    }
    return s;
}
// a - b + K p in (0, bound(a) + K p) for b < K p
template <int K>
ECG_HD Fp fp_sub_lazy_k(const Fp& a, const Fp& b) {
    constexpr FpConst kp = fp_p_times(K);
    Fp s;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        int32_t t = (int32_t)a.l[i] - (int32_t)b.l[i] + (int32_t)kp.l[i] + c;
        s.l[i] = i + 1 < FP_N ? ((u32)t & FP_MASK) : (u32)t;
        c = t >> 30;
    }
    return s;
}

// a - 2b mod p in [0, 2p) for a, b < 2p (X3 = E^2 - 2D of a point doubling): the difference, in (-4p, 2p), gets 4p added back
// when it is negative, then one conditional subtraction of 2p -- ~140 instructions against two modular operations' 208.
ECG_HD Fp fp_sub_dbl(const Fp& a, const Fp& b) {
    constexpr FpConst p4 = fp_p_times(4);
    Fp d;
    int32_t bw = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        int32_t t = (int32_t)a.l[i] - (int32_t)(b.l[i] << 1) + bw;
        d.l[i] = (u32)t & FP_MASK;
        bw = t >> 30;
    }
    const u32 m = (u32)bw;  // all ones: negative
    Fp r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        u32 t = d.l[i] + (p4.l[i] & m) + c;
        r.l[i] = t & FP_MASK;
        c = t >> 30;
    }
    return fp_cond_sub(r, blsc::P2);
}

// x < KIN p (limbs normalized, the top limb carrying the excess) -> the same residue below KOUT p, by conditional subtractions of
// KOUT 2^k p, k descending from the smallest k with KOUT 2^(k+1) >= KIN: after the step for C the value is below C.
template <int KIN, int KOUT>
ECG_HD Fp fp_reduce_below(const Fp& x) {
    if constexpr (KIN <= KOUT) {
        return x;
    } else {
        constexpr int k = KIN <= 2 * KOUT ? 0 : KIN <= 4 * KOUT ? 1 : KIN <= 8 * KOUT ? 2 : KIN <= 16 * KOUT ? 3 : -1;
        static_assert(k >= 0, "fp_reduce_below: more than four steps");
        constexpr FpConst c = fp_p_times(KOUT << k);
        return fp_reduce_below<(KOUT << k), KOUT>(fp_cond_sub(x, c.l));
    }
}
// 3t + 2z (S > 0) or 3t - 2z (S < 0) mod p, below KOUT p, for t < KT p and z < KZ p: the linear step that follows every Fp4
// squaring of a Granger-Scott / Karabina cyclotomic squaring.  As t + 2 (t +- z): one lazy addition / subtraction, one pass for
// the doubling and the sum, and the conditional subtractions -- against three modular operations (312 instructions) before.
template <int S, int KT, int KZ, int KOUT>
ECG_HD Fp fp_gs_lin(const Fp& t, const Fp& z) {
    Fp d;
    if constexpr (S > 0) d = fp_add_lazy(t, z);  // < (KT + KZ) p
    else d = fp_sub_lazy_k<KZ>(t, z);            // t - z + KZ p < (KT + KZ) p
    Fp r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        const u32 v = (d.l[i] << 1) + t.l[i] + c;
        r.l[i] = i + 1 < FP_N ? (v & FP_MASK) : v;
        c = v >> 30;
    }
    return fp_reduce_below<KT + 2 * (KT + KZ), KOUT>(r);
}

// Sum of N products with ONE Montgomery reduction: (a_0 b_0 + ... + a_{N-1} b_{N-1}) / R mod p, result < 2p whenever
// the integer sum is < R p = 632 p^2 (operands are lazy sums / lazy negations; every caller states its bound).
// This is how the tower spends multiplier time instead of linear operations: on gfx950 a 104-instruction modular
// subtraction costs a sixth of a whole product, so Karatsuba (3 reductions + 5 linear operations per Fp2 product)
// loses to schoolbook with the signs folded into operands (4 half-products + 2 reductions, no linear operation on a
// result).  169 N + 169 multiply-adds + 13 quotient digits; the 64-bit columns absorb 15 products of 30-bit limbs, so
// they are renormalized every floor(15 / (N + 1)) rows.
template <int N, int I>
ECG_HD void fp_sumprod_row(u64* T, const Fp (&a)[N], const Fp (&b)[N]) {
    constexpr int ROWS = 15 / (N + 1);
    // column I + 12 is new to row I unless the carry-save pass after row I - 1 has just opened it
    constexpr bool after_pass = I > 0 && I % ROWS == 0;
    constexpr int FRESH = I == 0 ? 0 : (after_pass ? 13 : 12);
    ecg_mad_row<FRESH>(T + I, a[0].l, b[0].l[I]);
#pragma unroll
    for (int k = 1; k < N; k++) ecg_mad_row<13>(T + I, a[k].l, b[k].l[I]);
    const u32 m = ((u32)T[I] * blsc::N0) & FP_MASK;
    ecg_mad_row_p(T + I, m);
    T[I + 1] += T[I] >> 30;  // T[I] == 0 mod 2^30 now
    if constexpr ((I + 1) % ROWS == 0 && I + 1 < FP_N) {
        // carry-save pass over the 12 live columns, top down (each step reads a high dword no earlier step has touched);
        // it opens column I + 13
        ecg_col_pass_hi<true>(T[I + 13], T[I + 12]);
#pragma unroll
        for (int c = I + 11; c >= I + 1; c--) ecg_col_pass_hi<false>(T[c + 1], T[c]);
    }
    if constexpr (I + 1 < FP_N) fp_sumprod_row<N, I + 1>(T, a, b);
}
template <int N>
ECG_HD Fp fp_sumprod(const Fp (&a)[N], const Fp (&b)[N]) {
    constexpr int ROWS = 15 / (N + 1);
    static_assert(ROWS >= 1, "too many products per row for the 64-bit columns");
    ECG_COUNT_MAD(169 * N + 182);
    ECG_SCHED_FENCE();  // one sum at a time: interleaving independent sums multiplies the live column accumulators
    u64 T[27];
    fp_sumprod_row<N, 0>(T, a, b);
    T[25] = 0;
    Fp r;
#pragma unroll
    for (int c = 13; c < 25; c++) {
        T[c + 1] += T[c] >> 30;
        r.l[c - 13] = (u32)T[c] & FP_MASK;
    }
    r.l[12] = (u32)T[25];
    ECG_SCHED_FENCE();
    return r;
}
ECG_HD Fp fp_sumprod2(const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1) {
    const Fp a[2] = {a0, a1};
    const Fp b[2] = {b0, b1};
    return fp_sumprod<2>(a, b);
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
ECG_HD Fp fp_sqr_body(const Fp& a) {
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
    // carry-save pass before the reduction rows add 13 more products per column (top down: ecg_col_pass_hi)
#pragma unroll
    for (int c = 24; c >= 0; c--) ecg_col_pass_hi<false>(T[c + 1], T[c]);
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
ECG_HD_NOINLINE Fp fp_sqr(Fp a) {
    ECG_COUNT_SQR();
    return fp_sqr_body(a);
}

// small-constant multiples
ECG_HD Fp fp_mul3(const Fp& a) { return fp_add(fp_dbl(a), a); }

// a^((p-3)/4), the one exponent the pipeline raises to (square roots and inverse square roots: fp_sqrt_inv,
// fp2_sqrt_with_norm_root): sliding windows of up to 5 bits over the odd powers a, a^3 .. a^31 along the generated
// schedule blsc::POW_PM3D4_SCHED -- 376 squarings + 81 products (the 4-bit fixed window took 376 + 104).  The window's table
// entry is read BEFORE the squarings that precede its use, and those are inlined: the table lives in the private segment
// (dynamic index), a load issued right in front of the product was an exposed trip to memory per window, and every
// out-of-line routine starts by waiting for ALL outstanding memory operations (the calling convention's s_waitcnt 0), so a
// load cannot be hidden behind a call.
ECG_HD_NOINLINE Fp fp_pow_pm3d4(Fp a) {
    Fp tab[16];
    tab[0] = a;
    const Fp a2 = fp_sqr(a);
    for (int i = 1; i < 16; i++) tab[i] = fp_mul(tab[i - 1], a2);
    Fp r = ecg_priv_load(tab[blsc::POW_PM3D4_SCHED[0][1]]);
    for (int w = 1; w < blsc::POW_PM3D4_STEPS; w++) {
        const u32 nsq = blsc::POW_PM3D4_SCHED[w][0], k = blsc::POW_PM3D4_SCHED[w][1];
        const Fp m = ecg_priv_load(tab[k & 15]);
        for (u32 q = 0; q < nsq; q++) {
            ECG_COUNT_SQR();
            r = fp_sqr_body(r);  // inlined: a call would wait for the load above at its entry (the ABI's s_waitcnt 0)
        }
        if (k != 255) r = fp_mul(r, m);
    }
    return r;
}

// ---- inversion: Bernstein-Yang division steps ("safegcd", https://gcd.cr.yp.to/papers.html#safegcd) --------------------------
// Fermat inversion is a 381-bit exponentiation: 380 squarings + 95 products = ~260 000 instructions on the lane that runs it,
// 0.5 ms of a lone chain (message stage, affine conversions, the Fp12 inversion of the final exponentiation).  The division
// steps walk (f, g) = (p, a) to (+-1, 0) with shifts and additions: 30 batches of 30 steps on the low words (the half-delta
// variant needs at most (45907 * 381 + 26313) / 19929 = 879 steps for a 381-bit modulus), each batch a 2 x 2 transition
// matrix applied to f, g and -- modulo p -- to the Bezout coefficients d, e: ~30 000 instructions, the same for every lane
// (no data-dependent branch).  Limbs are the field's own 30 bits, signed, the top limb carrying the sign.
struct FpDivMatrix {
    int32_t u, v, q, r;
};
ECG_HD int32_t fp_divsteps30(int32_t zeta, u32 f0, u32 g0, FpDivMatrix& t) {
    u32 u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    for (int i = 0; i < 30; i++) {
        u32 c1 = (u32)(zeta >> 31);  // all ones: delta > 0 (zeta = -(delta + 1/2))
        const u32 c2 = 0u - (g & 1);
        const u32 x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // -f, -u, -v when delta > 0
        g += x & c2;
        q += y & c2;
        r += z & c2;
        c1 &= c2;                           // swap only when g is odd as well
        zeta = (int32_t)((u32)zeta ^ c1) - 1;
        f += g & c1;
        u += q & c1;
        v += r & c1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t.u = (int32_t)u;
    t.v = (int32_t)v;
    t.q = (int32_t)q;
    t.r = (int32_t)r;
    return zeta;
}
// (f, g) <- t (f, g) / 2^30, exact
ECG_HD void fp_div_update_fg(int32_t* f, int32_t* g, const FpDivMatrix& t) {
    int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0];
    int64_t cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < FP_N; i++) {
        cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
        cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
        f[i - 1] = (int32_t)cf & (int32_t)FP_MASK;
        g[i - 1] = (int32_t)cg & (int32_t)FP_MASK;
        cf >>= 30;
        cg >>= 30;
    }
    f[FP_N - 1] = (int32_t)cf;
    g[FP_N - 1] = (int32_t)cg;
}
// (d, e) <- t (d, e) / 2^30 mod p, values kept in (-2p, p): the multiple of p that makes the low 30 bits vanish is added first
ECG_HD void fp_div_update_de(int32_t* d, int32_t* e, const FpDivMatrix& t) {
    const int32_t sd = d[FP_N - 1] >> 31, se = e[FP_N - 1] >> 31;
    int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
    int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0];
    int64_t ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
    md -= (int32_t)((blsc::P_INV30 * (u32)cd + (u32)md) & FP_MASK);
    me -= (int32_t)((blsc::P_INV30 * (u32)ce + (u32)me) & FP_MASK);
    cd += (int64_t)blsc::P[0] * md;
    ce += (int64_t)blsc::P[0] * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < FP_N; i++) {
        cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + (int64_t)blsc::P[i] * md;
        ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + (int64_t)blsc::P[i] * me;
        d[i - 1] = (int32_t)cd & (int32_t)FP_MASK;
        e[i - 1] = (int32_t)ce & (int32_t)FP_MASK;
        cd >>= 30;
        ce >>= 30;
    }
    d[FP_N - 1] = (int32_t)cd;
    e[FP_N - 1] = (int32_t)ce;
}
// r <- +-(r + (p if add_p)), limbs renormalised (the top limb keeps the sign)
ECG_HD void fp_div_fix(int32_t* r, int32_t add_mask, int32_t neg_mask) {
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        int32_t w = r[i] + (int32_t)(blsc::P[i] & (u32)add_mask);
        w = (w ^ neg_mask) - neg_mask;
        w += c;
        if (i < FP_N - 1) {
            r[i] = w & (int32_t)FP_MASK;
            c = w >> 30;
        } else {
            r[i] = w;
        }
    }
}
ECG_HD_NOINLINE Fp fp_inv(const Fp& a_in) {  // 0 -> 0
    const Fp a = fp_canon(fp_mul(a_in, blsc::ONE));  // the Montgomery residue a R as an integer in [0, p), whatever lazy form came in
    int32_t f[FP_N], g[FP_N], d[FP_N], e[FP_N];
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        f[i] = (int32_t)blsc::P[i];
        g[i] = (int32_t)a.l[i];
        d[i] = 0;
        e[i] = 0;
    }
    e[0] = 1;
    int32_t zeta = -1;
    ECG_COUNT_MAD(30 * (52 + 80));  // multiply instructions of the 30 matrix applications (fg: 4 x 13, de: 6 x 13 + 2)
    for (int it = 0; it < 30; it++) {
        FpDivMatrix t;
        zeta = fp_divsteps30(zeta, (u32)f[0], (u32)g[0], t);
        fp_div_update_de(d, e, t);
        fp_div_update_fg(f, g, t);
    }
    // g = 0, f = +-gcd = +-1 (f = p when a = 0: d is 0 then), d = f / a mod p in (-2p, p)
    fp_div_fix(d, d[FP_N - 1] >> 31, f[FP_N - 1] >> 31);
    fp_div_fix(d, d[FP_N - 1] >> 31, 0);
    Fp raw;
#pragma unroll
    for (int i = 0; i < FP_N; i++) raw.l[i] = (u32)d[i];
    return fp_mul(raw, blsc::R3);  // (a R)^-1 R^3 / R = a^-1 R
}

// Square root for p = 3 mod 4.  Returns true and s with s^2 == a when a is a square.
// Also hands back t = a^((p-3)/4): when a is a non-zero square, t == 1/s.
ECG_HD bool fp_sqrt_inv(const Fp& a, Fp& s, Fp& inv_s) {
    inv_s = fp_pow_pm3d4(a);
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

#if defined(ECG_TOWER_CALLS)
// COMPACT-CODE variant: Karatsuba over three out-of-line Fp products.  Slower on a healthy box than the
// sums of products below (9 300 vs 6 700 cycles) but a few hundred bytes per use instead of 9 KB: on a box whose
// instruction fetch does not keep up beyond the 64 KB instruction cache it is the faster one.  Operand components may be
// lazy sums < 8p (inner sums < 16p, 256 p^2 < 632 p^2).
ECG_HD Fp2 fp2_mul(const Fp2& a, const Fp2& b) {
    Fp t0 = fp_mul(a.c0, b.c0);
    Fp t1 = fp_mul(a.c1, b.c1);
    Fp t2 = fp_mul(fp_add_lazy(a.c0, a.c1), fp_add_lazy(b.c0, b.c1));
    return Fp2{fp_sub(t0, t1), fp_sub(fp_sub(t2, t0), t1)};
}
// operand components < 2p
ECG_HD Fp2 fp2_sqr(const Fp2& a) {
    Fp t0 = fp_mul(fp_add_lazy(a.c0, a.c1), fp_sub_lazy(a.c0, a.c1));
    Fp t1 = fp_mul(a.c0, a.c1);
    return Fp2{t0, fp_dbl(t1)};
}
#else
// Two sums of two products, the minus sign of i^2 folded into a lazily negated operand: 4 half-products and 2
// reductions, no linear operation on a result (Karatsuba's 3 products cost 5 of them: measured 9300 vs 6700 cycles,
// profiles/r01zf_fpbench.txt).  Operand components may be lazy sums < 8p: each sum is below 2 * 8p * 8p = 128 p^2.
ECG_HD Fp2 fp2_mul(const Fp2& a, const Fp2& b) {
    const Fp nb1 = fp_neg_lazy<8>(b.c1);
    return Fp2{fp_sumprod2(a.c0, b.c0, a.c1, nb1), fp_sumprod2(a.c0, b.c1, a.c1, b.c0)};
}
// (a0 + a1)(a0 - a1) + (2 a0 a1) i: 2 Fp products.  Operand components < 4p (a0 - a1 + 4p < 8p, sums < 8p).
ECG_HD Fp2 fp2_sqr(const Fp2& a) {
    const Fp s[1] = {fp_add_lazy(a.c0, a.c1)}, d[1] = {fp_sub_lazy_k<4>(a.c0, a.c1)};
    const Fp x[1] = {a.c0}, y[1] = {fp_add_lazy(a.c1, a.c1)};
    return Fp2{fp_sumprod<1>(s, d), fp_sumprod<1>(x, y)};
}
#endif
// the same square for lazy components < K p (K <= 8: (a0 + a1)(a0 - a1 + K p) < 16p * 16p, a0 (2 a1) < 8p * 16p)
template <int K>
ECG_HD Fp2 fp2_sqr_lazy(const Fp2& a) {
    const Fp s[1] = {fp_add_lazy(a.c0, a.c1)}, d[1] = {fp_sub_lazy_k<K>(a.c0, a.c1)};
    const Fp x[1] = {a.c0}, y[1] = {fp_add_lazy(a.c1, a.c1)};
    return Fp2{fp_sumprod<1>(s, d), fp_sumprod<1>(x, y)};
}
ECG_HD Fp2 fp2_sub_dbl(const Fp2& a, const Fp2& b) { return Fp2{fp_sub_dbl(a.c0, b.c0), fp_sub_dbl(a.c1, b.c1)}; }
ECG_HD Fp2 fp2_inv(const Fp2& a) {
    Fp d = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
    return Fp2{fp_mul(a.c0, d), fp_neg(fp_mul(a.c1, d))};
}

// Sum of M Fp2 products with two reductions in all: sum_k x_k y_k, where ny[k] = K p - y_k.c1 is handed in (a lazy
// negation is shared by every sum the same y_k appears in).  Bound: the 2M half-products of each component must sum
// below 632 p^2.
template <int M>
ECG_HD Fp2 fp2_sumprod(const Fp2 (&x)[M], const Fp2 (&y)[M], const Fp (&ny)[M]) {
    Fp a[2 * M], br[2 * M], bi[2 * M];
#pragma unroll
    for (int k = 0; k < M; k++) {
        a[2 * k] = x[k].c0;
        a[2 * k + 1] = x[k].c1;
        br[2 * k] = y[k].c0;  // re: xr yr - xi yi
        br[2 * k + 1] = ny[k];
        bi[2 * k] = y[k].c1;  // im: xr yi + xi yr
        bi[2 * k + 1] = y[k].c0;
    }
    return Fp2{fp_sumprod<2 * M>(a, br), fp_sumprod<2 * M>(a, bi)};
}
// xi a = (a0 - a1) + (a0 + a1) i as a product operand: components < 2 K p for components of a < K p
template <int K>
ECG_HD Fp2 fp2_mul_xi_lazy(const Fp2& a) { return Fp2{fp_sub_lazy_k<K>(a.c0, a.c1), fp_add_lazy(a.c0, a.c1)}; }

// Square root in Fp2 by the norm ("complex") method, given s with s^2 = norm(a) = a0^2 + a1^2 and a1 != 0.
// ONE Fp exponentiation: with d = (a0 + s)/2, t = d^((p-3)/4) and c = t d, either c^2 = d (then x0 = c, 1/x0 = t) or
// c^2 = -d -- and then the OTHER candidate (a0 - s)/2 = a1^2 / (4 c^2) is the square, with root a1 / (2c) = -a1 t / 2
// (c t = d^((p-1)/2) = -1) and imaginary part a1 / (2 x0) = c.  No second exponentiation, no data-dependent branch
// around one: the lanes of a wave stay together.  Any root; false if r^2 != a.
ECG_HD bool fp2_sqrt_with_norm_root(const Fp2& a, const Fp& s, Fp2& r) {
    const Fp d = fp_mul(fp_add(a.c0, s), blsc::INV2);
    const Fp t = fp_pow_pm3d4(d);
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
