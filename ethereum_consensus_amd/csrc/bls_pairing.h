// Optimal-ate pairing check on BLS12-381 for the gfx950 BLS path: the e(pk, H(m)) == e(g1, sig)
// equation blst evaluates for /root/reference/ethereum-consensus/src/crypto/bls.rs:71,106,126.
//
// Miller loop over |x| = 0xd201000000010000 on the M-twist with inversion-free steps in homogeneous projective
// coordinates (x = X/Z, y = Y/Z); all pairs of one product share the accumulator squaring.  Line through the untwisted T
// evaluated at P = (xP, yP), times w^3 and times an Fp2 factor (both killed by the final exponentiation):
//     doubling:  (Y^2 - 3 b' Z^2) + (-3 X^2 xP) w^2 + (2 Y Z yP) w^3                                    b' = 4 xi
//     addition:  (theta xQ - lambda yQ) + (-theta xP) w^2 + (lambda yP) w^3    theta = Y - yQ Z, lambda = X - xQ Z
// i.e. the sparse shape fp12_mul_by_line multiplies by.  Final exponentiation: easy part, then
// 3 (p^4 - p^2 + 1)/r = (x-1)^2 (x+p) (x^2 + p^2 - 1) + 3 with Granger-Scott / Karabina cyclotomic squarings.
#pragma once
#include "bls_curve.h"

namespace ecg {

// ---- lane slots: twelve Fp values per lane in LDS ---------------------------------------------------------------------------
// The pairing lane kernels run ONE wave per SIMD (the whole 512-entry register file per lane), so a CU holds 256 lanes and
// its 160 KB of LDS give each of them 640 bytes: twelve field elements (624 B).  Round 2 kept the running points of the
// Miller loop and the operands of the out-of-line Fp12 routines in the private segment: 515 MB of frames for 65 536 lanes,
// which no cache holds, so every reload was a trip to HBM that a lone wave waits out (19.8 GB of traffic per launch, 20 % of
// the wave's cycles parked on s_waitcnt).  The slots hold, in turn, the two running points of the Miller loop (2 x 6 Fp) and
// one Fp12 operand of the final exponentiation.  Loads are volatile:
// the point of the slots is that a value is re-read where it is used instead of staying live (and spilling).
#ifndef ECG_LANE_SLOTS
#define ECG_LANE_SLOTS 12  // bls_pairing2_kernels.hip (two lanes per tuple, two waves per SIMD) compiles with 6: 320 bytes of LDS per lane
#endif
constexpr int LANE_SLOTS = ECG_LANE_SLOTS;
#if defined(__HIP_DEVICE_COMPILE__)
// limb j of slot s at dword [13 s + j][lane]: every address is lane * 4 + constant (ONE address register for the whole kernel),
// every access a ds_read_b32 / ds_write_b32 of 64 consecutive dwords (conflict-free).  Dword accesses on purpose: a
// ds_read_b128 delivers, and a ds_write_b128 demands, a 128-bit register TUPLE, the register coalescer then keeps the limbs
// that pass through one in tuples for their whole life, and a tuple can only be spilled and reloaded whole -- the first
// version of the slots had four limbs of the Miller accumulator reloaded from the private segment in front of over a hundred
// single-limb uses per line product.  (13 LDS instructions per field element instead of 4: +0.5 % instructions.)
// INVARIANT: a slot is indexed by threadIdx.x & 63, so a workgroup that reaches miller_loop / fp12_cyc_pow_x / fp12_mul_slots
// must be ONE wave (two waves of a 128-thread block would share slots and silently corrupt each other's pairings).  Every
// kernel that includes this header launches with ECG_LANE_SLOT_BLOCK threads; bls_kernels.h asserts BLS_BLOCK equals it.
#define ECG_LANE_SLOT_BLOCK 64
static __shared__ u32 g_lane_slots[13 * LANE_SLOTS * ECG_LANE_SLOT_BLOCK];
// explicit LDS pointer: a volatile access through a generic pointer compiles to flat_load / flat_store
typedef __attribute__((address_space(3))) volatile u32* lane_slot_ptr;
ECG_D Fp slot_load(int s) {
    const lane_slot_ptr p = (lane_slot_ptr)&g_lane_slots[13 * s * 64 + (threadIdx.x & 63)];
    Fp r;
#pragma unroll
    for (int j = 0; j < 13; j++) r.l[j] = p[64 * j];
    return r;
}
ECG_D void slot_store(int s, const Fp& a) {
    const lane_slot_ptr p = (lane_slot_ptr)&g_lane_slots[13 * s * 64 + (threadIdx.x & 63)];
#pragma unroll
    for (int j = 0; j < 13; j++) p[64 * j] = a.l[j];
}
#else
// host lane simulator: one lane at a time per thread
static thread_local Fp g_lane_slots[LANE_SLOTS];
ECG_HD Fp slot_load(int s) { return g_lane_slots[s]; }
ECG_HD void slot_store(int s, const Fp& a) { g_lane_slots[s] = a; }
#endif
ECG_HD Fp2 slot_load2(int s) { return Fp2{slot_load(s), slot_load(s + 1)}; }
ECG_HD void slot_store2(int s, const Fp2& a) {
    slot_store(s, a.c0);
    slot_store(s + 1, a.c1);
}
// running point of Miller pair k: slots 6k .. 6k+5 = X, Y, Z
ECG_HD J2 slot_load_point(int k) { return J2{slot_load2(6 * k), slot_load2(6 * k + 2), slot_load2(6 * k + 4)}; }
ECG_HD void slot_store_point(int k, const J2& t) {
    slot_store2(6 * k, t.x);
    slot_store2(6 * k + 2, t.y);
    slot_store2(6 * k + 4, t.z);
}
ECG_HD Fp12 slot_load_fp12() {
    Fp12 r;
    r.c0.c0 = slot_load2(0);
    r.c0.c1 = slot_load2(2);
    r.c0.c2 = slot_load2(4);
    r.c1.c0 = slot_load2(6);
    r.c1.c1 = slot_load2(8);
    r.c1.c2 = slot_load2(10);
    return r;
}
ECG_HD void slot_store_fp12(const Fp12& a) {
    slot_store2(0, a.c0.c0);
    slot_store2(2, a.c0.c1);
    slot_store2(4, a.c0.c2);
    slot_store2(6, a.c1.c0);
    slot_store2(8, a.c1.c1);
    slot_store2(10, a.c1.c2);
}

struct MillerPair {
    Fp px, py;  // P in E1 affine
    Fp npx;     // 2p - px: the minus sign of the line's w^2 coefficient as a product operand
    Fp n3px;    // 3 (2p - px) (lazy, < 6p): the doubling step's
    Fp2 qx, qy; // Q in E2 affine
    J2 t;       // running point (HOMOGENEOUS projective: x = X/Z, y = Y/Z): the caller's initial value; miller_loop keeps the live one in the lane slots
    u32 active; // 0: P or Q is infinity, the pair contributes 1
};

ECG_HD void miller_pair_init(MillerPair& m, const A1& p, const A2& q) {
    m.active = (p.inf || q.inf) ? 0u : 1u;
    m.px = p.x;
    m.py = p.y;
    m.npx = fp_neg_lazy<2>(p.x);
    m.n3px = fp_add_lazy(fp_add_lazy(m.npx, m.npx), m.npx);
    m.qx = q.x;
    m.qy = q.y;
    m.t.x = q.x;
    m.t.y = q.y;
    m.t.z = fp2_one();
}

// T <- 2T, f <- f * line_{T,T}(P); T = running point of pair k, read from the lane slots where it is used.
// Homogeneous projective coordinates on the twist E': y^2 = x^3 + b', b' = 4 xi (Costello-Lange-Naehrig 2010; Aranha et al.
// 2011, eq. 10, scaled by 4 so that nothing is halved):
//   B = Y^2, J = X^2, H = 2 Y Z, E = 3 b' Z^2 = 3 xi (2Z)^2 (brought below 2p: an operand of the four terms below), F = 3E,
//   X3 = (2 X Y)(B - F),   Y3 = (B + F)^2 - 12 E^2 = B (B + 6E) + E (-3E),   Z3 = (4B) H,
//   line (the tangent scaled by Z^2; any Fp2 factor dies in the final exponentiation): l0 = B - E, l1 = -3 J xP, l2 = H yP.
// 3 squarings, 4 products, 1 sum of two products, 4 Fp products over lazy operands.  (Jacobian coordinates, until late in
// round 3: 4 squarings, 5 products, 1 sum of two products, 4 Fp products -- the Jacobian line needs E Z^2 and Z3 Z^2 as
// products of their own.)
ECG_MILLER_DBL_FN void miller_dbl_step(Fp12& f, const MillerPair& m, int k) {
    const int sx = 6 * k, sy = sx + 2, sz = sx + 4;
    Fp2 H, E;
    {
        const Fp2 Z = slot_load2(sz);
        {
            const Fp2 Y = slot_load2(sy);
            H = fp2_mulx(fp2_add_lazy(Y, Y), Z);
        }
        const Fp2 xc = fp2_mul_xi_lazy<2>(fp2_sqr_lazy<4>(fp2_add_lazy(Z, Z)));  // xi (2Z)^2, components < 4p
        E.c0 = fp_reduce_below<12, 2>(fp_add_lazy(fp_add_lazy(xc.c0, xc.c0), xc.c0));
        E.c1 = fp_reduce_below<12, 2>(fp_add_lazy(fp_add_lazy(xc.c1, xc.c1), xc.c1));
    }
    Fp2 B, XY2, l1;
    {
        const Fp2 Y = slot_load2(sy);
        B = fp2_sqrx(Y);
        const Fp2 X = slot_load2(sx);
        XY2 = fp2_mulx(fp2_add_lazy(X, X), Y);
        l1 = fp2_mul_fp(fp2_sqrx(X), m.n3px);
    }
    const Fp2 l2 = fp2_mul_fp(H, m.py);
    {
        const Fp2 B2 = fp2_add_lazy(B, B);
        slot_store2(sz, fp2_mulx(fp2_add_lazy(B2, B2), H));  // Z3 = (4B) H
    }
    const Fp2 l0 = f_sub_lazy<2>(B, E);  // < 4p
    {
        const Fp2 F = fp2_add_lazy(fp2_add_lazy(E, E), E);  // 3E < 6p
        slot_store2(sx, fp2_mulx(XY2, f_sub_lazy<6>(B, F)));  // X3 = (2XY)(B - F + 6p)
        // Y3 = B (B + 6E) + E (6p - 3E): bounds (units of p^2 per coefficient, Fp2 doubles them) 2 * 14 + 2 * 6 = 40
        slot_store2(sy, f_sp2<14, 6>(B, fp2_add_lazy(fp2_add_lazy(B, F), F), E, f_neg_lazy<6>(F)));
    }
    fp12_mul_by_line<4>(f, l0, l1, l2);
}

// T <- T + Q (Q affine), f <- f * line_{T,Q}(P): the mixed addition in the same coordinates (madd-1998-cmo with both signs turned),
//   theta = Y - yQ Z, lambda = X - xQ Z, C = theta^2, D = lambda^2, E = lambda D, F = Z C, G = X D, A = E + F - 2G,
//   X3 = lambda A, Y3 = theta (G - A) - E Y, Z3 = Z E,   line: l0 = theta xQ - lambda yQ, l1 = -theta xP, l2 = lambda yP.
// Five steps per pair and check: modular linear operations as they come.
ECG_HD void miller_add_step_inl(Fp12& f, const MillerPair& m, int k) {
    const J2 T = slot_load_point(k);
    const Fp2 th = fp2_sub(T.y, fp2_mulx(m.qy, T.z));
    const Fp2 la = fp2_sub(T.x, fp2_mulx(m.qx, T.z));
    const Fp2 D = fp2_sqrx(la);
    const Fp2 E = fp2_mulx(la, D);
    const Fp2 F = fp2_mulx(T.z, fp2_sqrx(th));
    const Fp2 G = fp2_mulx(T.x, D);
    const Fp2 A = fp2_sub(fp2_add(E, F), fp2_dbl(G));
    const Fp2 X3 = fp2_mulx(la, A);
    const Fp2 Y3 = fp2_sub(fp2_mulx(th, fp2_sub(G, A)), fp2_mulx(E, T.y));
    const Fp2 Z3 = fp2_mulx(T.z, E);
    const Fp2 l0 = fp2_sub(fp2_mulx(th, m.qx), fp2_mulx(la, m.qy));
    const Fp2 l1 = fp2_mul_fp(th, m.npx);
    const Fp2 l2 = fp2_mul_fp(la, m.py);
    slot_store_point(k, J2{X3, Y3, Z3});
    fp12_mul_by_line<2>(f, l0, l1, l2);
}

// f = prod_k f_{|x|,Q_k}(P_k), conjugated (x < 0)
// The accumulator is a local VALUE whose address never leaves this function: it lives in VGPRs/AGPRs across the whole loop.
// The five addition steps are inlined as well (round 2 ran them out of line on a copy): a copy of the accumulator to or
// from the private segment is a merged dwordx4 access, and the register coalescer then keeps the accumulator's limbs in
// 128-bit tuples for the WHOLE loop, which the allocator can only spill and reload four at a time (see common.h).  The
// running points live in the lane slots (LDS), the fixed coordinates of the pairs in the private segment (read once per
// step, long before their use).
ECG_HD_NOINLINE void miller_loop(Fp12& f, MillerPair* pairs, int n) {
    ECG_LONG_BRANCH_GUARD();  // loops far larger than the branch range: see common.h
    bool any = false;
    MillerPair lp[2];  // n <= 2 (verify: 2 pairs; aggregate_verify: 1 per lane); private copies, see ecg_priv_load
    for (int k = 0; k < n && k < 2; k++) {
        lp[k] = ecg_priv_load(pairs[k]);
        any = any || lp[k].active;
        slot_store_point(k, lp[k].t);
    }
    Fp12 acc;
    fp12_set_one(acc);
    if (any) {
        for (int b = 62; b >= 0; b--) {
            if (b != 62) fp12_sqr(acc, acc);
            for (int k = 0; k < n; k++)
                if (lp[k].active) miller_dbl_step(acc, lp[k], k);
            if ((blsc::X_ABS >> b) & 1)
                for (int k = 0; k < n; k++)
                    if (lp[k].active) {
                        miller_add_step_inl(acc, lp[k], k);
                    }
        }
        fp12_conj(acc, acc);
    }
    ecg_priv_store(f, acc);
}

// r = a * (the Fp12 value in the lane slots): Karatsuba over three Fp6 products, the second operand read from LDS where it is
// used.  r may alias a.
ECG_HD Fp6 slot_load_fp6(int s) { return Fp6{slot_load2(s), slot_load2(s + 2), slot_load2(s + 4)}; }
ECG_HD void fp12_mul_by_slots_inl(Fp12& r, const Fp12& a) {
    Fp6 t0, t1, m;
    {
        const Fp6 b0 = slot_load_fp6(0);
        fp6_mul(t0, a.c0, b0);
    }
    {
        const Fp6 b1 = slot_load_fp6(6);
        fp6_mul(t1, a.c1, b1);
    }
    {
        const Fp6 b0 = slot_load_fp6(0), b1 = slot_load_fp6(6);
        fp6_mul_sums(m, a.c0, a.c1, b0, b1);
    }
    fp12_karatsuba_combine(r.c0, r.c1, m, t0, t1);
}

// r = a * b for two values in the caller's private segment, the second operand through the lane slots: the out-of-line
// product of bls_tower.h holds both operands and its Fp6 temporaries in registers and spills (217 scratch instructions per
// call); this one reads b from LDS coefficient by coefficient, as the exponentiations by x do.  The slots must be free (they
// are between two exponentiations).  r may alias a or b.
ECG_HD_NOINLINE void fp12_mul_slots(Fp12& r, const Fp12& a, const Fp12& b) {
    {
        const Fp12 y = ecg_priv_load(b);
        slot_store_fp12(y);
    }
    Fp12 x = ecg_priv_load(a);
    fp12_mul_by_slots_inl(x, x);
    ecg_priv_store(r, x);
}

// a^x for a in the cyclotomic subgroup (x < 0: conjugate)
ECG_HD_NOINLINE void fp12_cyc_pow_x(Fp12& r, const Fp12& a) {
    // The running value is register-resident (squaring and product are inlined) and never passes through memory inside the
    // loop: a copy to or from the private segment is a merged dwordx4 access, the register coalescer then keeps the limbs
    // in 128-bit tuples for the whole loop and the allocator spills and reloads them four at a time (round 2: 9 exposed
    // reloads per squaring).  The BASE is needed five times in 63 iterations; it waits in the lane slots (LDS) -- 156 more
    // live dwords under the squaring would come back as spills -- and the product reads it from there, coefficient by
    // coefficient.  `a` may alias `r` (written only at the end).
    ECG_LONG_BRANCH_GUARD();  // the loops below are hundreds of KB of code: see common.h
    Fp12 acc = ecg_priv_load(a);
    slot_store_fp12(acc);
    // |x| = 0xd201000000010000: bits 63, 62, 60, 57, 48, 16.  MSB first: after bit 63 (acc = a) the squarings come in runs of
    // 1, 2, 3, 9, 32, 16 with a product by the base after each run but the last.  The two long runs go through Karabina's
    // compressed squaring (bls_tower.h: two Fp4 squarings instead of three, one decompression at the end of the run): -14 % of an
    // exponentiation's instructions; the short ones would not pay for their decompression.
    struct Run {
        u8 n, compressed;
    };
    ECG_CONST Run RUNS[6] = {{1, 0}, {2, 0}, {3, 0}, {9, 0}, {32, 1}, {16, 1}};
    for (int s = 0; s < 6; s++) {
        const u32 n = RUNS[s].n;
        if (RUNS[s].compressed) {
            Fp2 z2 = acc.c1.c0, z3 = acc.c0.c2, z4 = acc.c0.c1, z5 = acc.c1.c2;
            for (u32 k = 0; k < n; k++) fp12_cyclotomic_sqr_compressed(z2, z3, z4, z5);
            z2 = fp2_below_2p(z2);  // the run kept its coefficients below 4p (bls_tower.h)
            z3 = fp2_below_2p(z3);
            z4 = fp2_below_2p(z4);
            z5 = fp2_below_2p(z5);
            Fp2 z0, z1;
            fp12_cyclotomic_decompress(z0, z1, z2, z3, z4, z5);
            acc.c0.c0 = z0;
            acc.c0.c1 = z4;
            acc.c0.c2 = z3;
            acc.c1.c0 = z2;
            acc.c1.c1 = z1;
            acc.c1.c2 = z5;
        } else {
            for (u32 k = 0; k < n; k++) fp12_cyclotomic_sqr_run(acc, acc);
            fp12_cyc_normalize(acc);
        }
        if (s < 5) fp12_mul_by_slots_inl(acc, acc);
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
    fp12_mul_slots(t, t, u);
    fp12_frob(u, t);
    fp12_frob(u, u);
    fp12_mul_slots(t, u, t);
    // hard part
    fp12_cyc_pow_x(a, t);
    fp12_conj(u, t);
    fp12_mul_slots(a, a, u);  // t^(x-1)
    fp12_cyc_pow_x(b, a);
    fp12_conj(u, a);
    fp12_mul_slots(a, b, u);  // t^((x-1)^2)
    fp12_cyc_pow_x(b, a);
    fp12_frob(u, a);
    fp12_mul_slots(b, b, u);  // a^(x+p)
    fp12_cyc_pow_x(c, b);
    fp12_cyc_pow_x(c, c);
    fp12_frob(u, b);
    fp12_frob(u, u);
    fp12_mul_slots(c, c, u);
    fp12_conj(u, b);
    fp12_mul_slots(c, c, u);  // b^(x^2 + p^2 - 1)
    fp12_cyclotomic_sqr(u, t);
    fp12_mul_slots(u, u, t);  // t^3
    fp12_mul_slots(c, c, u);
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
