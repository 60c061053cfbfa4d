// hash_to_curve for G2, suite BLS12381G2_XMD:SHA-256_SSWU_RO_ with the Ethereum proof-of-possession
// DST (/root/reference/ethereum-consensus/src/crypto/bls.rs:22 `BLS_DST`; performed inside blst for
// every verify/sign call, crypto/bls.rs:71,106,126,218).  RFC 9380: expand_message_xmd ->
// hash_to_field (2 x Fp2) -> simplified SWU on E2' -> 3-isogeny -> add -> clear cofactor
// (Budroni-Pintore via psi).  One lane hashes one message.
#pragma once
#include "bls_curve.h"
#include "sha256.h"

namespace ecg {

namespace blsc {
// DST || I2OSP(len(DST), 1)
ECG_CONST u8 DST_PRIME[44] = {'B', 'L', 'S', '_', 'S', 'I', 'G', '_', 'B', 'L', 'S', '1', '2', '3', '8',
                              '1', 'G', '2', '_', 'X', 'M', 'D', ':', 'S', 'H', 'A', '-', '2', '5', '6',
                              '_', 'S', 'S', 'W', 'U', '_', 'R', 'O', '_', 'P', 'O', 'P', '_', 43};
}  // namespace blsc

ECG_HD_NOINLINE void sha256_block(u32* st, u32* w) { sha256_compress(st, w); }

// byte-oriented SHA-256 with the block kept in the lane's private memory
struct Sha256B {
    u32 st[8];
    u8 blk[64];
    u32 fill;
    u64 total;
};
ECG_HD void shab_init(Sha256B& s) {
    for (int i = 0; i < 8; i++) s.st[i] = SHA256_IV[i];
    s.fill = 0;
    s.total = 0;
}
ECG_HD void shab_flush(Sha256B& s) {
    u32 w[16];
    for (int i = 0; i < 16; i++)
        w[i] = ((u32)s.blk[4 * i] << 24) | ((u32)s.blk[4 * i + 1] << 16) | ((u32)s.blk[4 * i + 2] << 8) | s.blk[4 * i + 3];
    sha256_block(s.st, w);
    s.fill = 0;
}
ECG_HD void shab_put(Sha256B& s, u8 b) {
    s.blk[s.fill++] = b;
    s.total++;
    if (s.fill == 64) shab_flush(s);
}
ECG_HD void shab_update(Sha256B& s, const u8* p, size_t n) {
    for (size_t i = 0; i < n; i++) shab_put(s, p[i]);
}
ECG_HD void shab_final(Sha256B& s, u8 out[32]) {
    const u64 bits = s.total * 8;
    shab_put(s, 0x80);
    while (s.fill != 56) shab_put(s, 0);
    for (int i = 7; i >= 0; i--) shab_put(s, (u8)(bits >> (8 * i)));
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (u8)(s.st[i] >> 24);
        out[4 * i + 1] = (u8)(s.st[i] >> 16);
        out[4 * i + 2] = (u8)(s.st[i] >> 8);
        out[4 * i + 3] = (u8)s.st[i];
    }
}

// expand_message_xmd(msg, DST, 256): 3 + 8 x 2 compressions for a 32-byte message
ECG_HD_NOINLINE void xmd_expand_256(u8* out, const u8* msg, size_t msg_len) {
    Sha256B s;
    u8 b0[32], bi[32];
    shab_init(s);
    for (int i = 0; i < 64; i++) shab_put(s, 0);  // Z_pad
    shab_update(s, msg, msg_len);
    shab_put(s, 0x01);  // l_i_b_str = I2OSP(256, 2)
    shab_put(s, 0x00);
    shab_put(s, 0x00);  // I2OSP(0, 1)
    for (int i = 0; i < 44; i++) shab_put(s, blsc::DST_PRIME[i]);
    shab_final(s, b0);
    for (int k = 1; k <= 8; k++) {
        shab_init(s);
        for (int i = 0; i < 32; i++) shab_put(s, k == 1 ? b0[i] : (u8)(b0[i] ^ bi[i]));
        shab_put(s, (u8)k);
        for (int i = 0; i < 44; i++) shab_put(s, blsc::DST_PRIME[i]);
        shab_final(s, bi);
        for (int i = 0; i < 32; i++) out[32 * (k - 1) + i] = bi[i];
    }
}

// OS2IP(64 bytes) mod p -> Montgomery
ECG_HD Fp fp_from_be64(const u8* b) {
    u32 w[12];
    for (int i = 0; i < 12; i++) w[i] = 0;
    for (int i = 0; i < 4; i++) {
        const u8* q = b + 4 * (3 - i);
        w[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | q[3];
    }
    Fp hi = raw_from_words(w);
    Fp lo = raw_from_be48(b + 16, false);
    return fp_add(fp_mul(lo, blsc::R2), fp_mul(hi, blsc::R2_384));
}

ECG_HD Fp2 fp2_horner(const Fp2* c, int deg, const Fp2& x) {
    Fp2 acc = c[deg];
    for (int i = deg - 1; i >= 0; i--) acc = fp2_add(fp2_mulx(acc, x), c[i]);
    return acc;
}

// simplified SWU onto E2' (RFC 9380 6.6.2, straight-line form) then the 3-isogeny to E2, result in
// Jacobian coordinates (no inversion for the isogeny denominators).
// tv2 = Z^2 u^4 + Z u^2, the quantity SSWU inverts
ECG_HD Fp2 sswu_tv2(const Fp2& u) {
    const Fp2 tv1 = fp2_mulx(blsc::SSWU_Z, fp2_sqrx(u));
    return fp2_add(fp2_sqrx(tv1), tv1);
}
// tv2_inv_in: 1 / tv2 (ignored when tv2 = 0) -- the caller inverts the tv2 of both field elements of a message with ONE
// exponentiation (Montgomery's trick), see hash_to_g2.
ECG_HD_NOINLINE void map_to_curve_g2(J2& r_out, const Fp2& u_in, const Fp2& tv2_inv_in) {
    const Fp2 u = ecg_priv_load(u_in);  // operands are locals of the caller (private segment)
    const Fp2 tv2_inv = ecg_priv_load(tv2_inv_in);
    J2 r;
    Fp2 tv1 = fp2_mulx(blsc::SSWU_Z, fp2_sqrx(u));
    Fp2 tv2 = fp2_add(fp2_sqrx(tv1), tv1);
    Fp2 x1;
    if (fp2_is_zero(tv2)) {
        x1 = blsc::SSWU_B_OVER_ZA;
    } else {
        x1 = fp2_mulx(blsc::SSWU_MB_OVER_A, fp2_add(fp2_one(), tv2_inv));
    }
    Fp2 gx1 = fp2_add(fp2_add(fp2_mulx(fp2_sqrx(x1), x1), fp2_mulx(blsc::SSWU_A, x1)), blsc::SSWU_B);
    // Exactly one of gx1, gx2 = g(Z u^2 x1) = (Z u^2)^3 gx1 is a square.  Decide on the NORM of gx1 (one Fp
    // exponentiation, which is also the norm root a square gx1 needs), derive the norm root of gx2 from it when gx1 is
    // not a square -- norm(gx2) = m^3 n1 with m = norm(Z u^2) = 5 norm(u)^2; n1 is then a non-residue, s^2 = -n1, and
    // v = sqrt(-5) norm(u) has v^2 = -m (a constant: -5 is a residue, 5 and -1 are not), so (m s v)^2 = m^3 n1 -- and
    // take ONE Fp2 root of the chosen value.  Every lane of a wave runs the same 2 exponentiations, whichever of gx1, gx2
    // its message lands on, instead of up to 6 on divergent paths.
    const Fp2 x2 = fp2_mulx(tv1, x1);
    const Fp2 gx2 = fp2_add(fp2_add(fp2_mulx(fp2_sqrx(x2), x2), fp2_mulx(blsc::SSWU_A, x2)), blsc::SSWU_B);
    const Fp n1 = fp_add(fp_sqr(gx1.c0), fp_sqr(gx1.c1));
    Fp sn;
    const bool sq1 = fp_sqrt(n1, sn);  // sn = n1^((p+1)/4) either way
    if (!sq1) {
        const Fp m = fp_add(fp_sqr(tv1.c0), fp_sqr(tv1.c1));
        const Fp v = fp_mul(blsc::SQRT_M5, fp_add(fp_sqr(u.c0), fp_sqr(u.c1)));
        sn = fp_mul(fp_mul(m, sn), v);
    }
    Fp2 x = sq1 ? x1 : x2, y;
    const Fp2 g = sq1 ? gx1 : gx2;
    bool ok = !fp_is_zero(g.c1) && fp2_sqrt_with_norm_root(g, sn, y);
    if (!ok) {
        // real g (a1 = 0) or an input outside the theorem's assumptions: the general routine decides
        x = x1;
        if (!fp2_sqrt(gx1, y)) {
            x = x2;
            (void)fp2_sqrt(gx2, y);
        }
    }
    if (fp2_sgn0(u) != fp2_sgn0(y)) y = fp2_neg(y);
    // iso3: x' = xn/xd, y' = y yn/yd  ->  Jacobian with Z = xd yd
    Fp2 xn = fp2_horner(blsc::ISO_XNUM, 3, x);
    Fp2 xd = fp2_horner(blsc::ISO_XDEN, 2, x);
    Fp2 yn = fp2_horner(blsc::ISO_YNUM, 3, x);
    Fp2 yd = fp2_horner(blsc::ISO_YDEN, 3, x);
    if (fp2_is_zero(xd) || fp2_is_zero(yd)) {
        jac_set_inf(r);  // exceptional point of the isogeny
        ecg_priv_store(r_out, r);
        return;
    }
    Fp2 z = fp2_mulx(xd, yd);
    Fp2 yd2 = fp2_sqrx(yd);
    r.x = fp2_mulx(fp2_mulx(xn, xd), yd2);                                       // xn xd yd^2
    r.y = fp2_mulx(fp2_mulx(fp2_mulx(y, yn), fp2_mulx(fp2_sqrx(xd), xd)), yd2);  // y yn xd^3 yd^2
    r.z = z;
    ecg_priv_store(r_out, r);
}

// h_eff multiplication by Budroni-Pintore: [x^2 - x - 1] P + [x - 1] psi(P) + psi^2(2P)
ECG_HD_NOINLINE void g2_clear_cofactor(J2& r, const J2& p_in) {
    const J2 p = ecg_priv_load(p_in);
    J2 t1, t2, t3, n;
    jac_mul_xabs(t1, p);
    jac_neg(t1, t1);  // [x] P
    g2_psi(t2, p);    // psi(P)
    jac_dbl(t3, p);
    g2_psi(t3, t3);
    g2_psi(t3, t3);  // psi^2(2P)
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

ECG_HD_NOINLINE void hash_to_g2(A2& r, const u8* msg, size_t msg_len) {
    u8 xm[256];
    xmd_expand_256(xm, msg, msg_len);
    Fp2 u0 = Fp2{fp_from_be64(xm), fp_from_be64(xm + 64)};
    Fp2 u1 = Fp2{fp_from_be64(xm + 128), fp_from_be64(xm + 192)};
    // one inversion for the two SSWU maps: 1/t0 = t1 / (t0 t1), 1/t1 = t0 / (t0 t1); a zero tv2 (the exceptional case
    // of the map, which then ignores its inverse) is replaced by 1 so that it does not poison the other one
    Fp2 t0 = sswu_tv2(u0), t1 = sswu_tv2(u1);
    if (fp2_is_zero(t0)) t0 = fp2_one();
    if (fp2_is_zero(t1)) t1 = fp2_one();
    const Fp2 ti = fp2_inv(fp2_mulx(t0, t1));
    const Fp2 i0 = fp2_mulx(ti, t1), i1 = fp2_mulx(ti, t0);
    J2 q0, q1;
    map_to_curve_g2(q0, u0, i0);
    map_to_curve_g2(q1, u1, i1);
    jac_add(q0, q0, q1);
    g2_clear_cofactor(q0, q0);
    A2 a;
    jac_to_aff(a, q0);
    ecg_priv_store(r, a);
}

// hash_to_g2 for TWO lanes per message (small batches are all latency, bls.hip): lane j maps field element u_j of the message
// to the curve -- each lane inverts its own tv2 instead of sharing one inversion -- and one lane adds the two points, clears the
// cofactor and converts.  The two maps are ~45 % of the multiplies of hash_to_g2 and the only part with parallelism.
ECG_HD_NOINLINE void hash_to_g2_map(J2& q, const u8* msg, size_t msg_len, int j) {
    u8 xm[256];
    xmd_expand_256(xm, msg, msg_len);
    const Fp2 u = Fp2{fp_from_be64(xm + 128 * j), fp_from_be64(xm + 128 * j + 64)};
    Fp2 t = sswu_tv2(u);
    if (fp2_is_zero(t)) t = fp2_one();  // the exceptional case of the map ignores the inverse
    const Fp2 ti = fp2_inv(t);
    map_to_curve_g2(q, u, ti);
}
ECG_HD_NOINLINE void hash_to_g2_finish(A2& r, const J2& q0_in, const J2& q1_in) {
    J2 q0 = ecg_priv_load(q0_in);
    const J2 q1 = ecg_priv_load(q1_in);
    jac_add(q0, q0, q1);
    g2_clear_cofactor(q0, q0);
    A2 a;
    jac_to_aff(a, q0);
    ecg_priv_store(r, a);
}

}  // namespace ecg
