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

// ---- expand_message_xmd for the 32-byte messages every caller of this crate signs (signing roots: signing.rs:14-22) ------------
// Word-oriented and register-resident.  The byte-oriented routine above keeps its block in the private segment (one
// scratch_store_byte per message byte, 64 byte loads per block): 430 us per message at one wave per SIMD, five times what its
// instructions cost, 7 % of the message stage (profiles/r04f_h2c_parts.txt).  With msg_len = 32 every block boundary is known:
//   b_0 = H(Z_pad | msg | 0x0100 | 0x00 | DST')   = [64 zero bytes] [msg, 01 00 00, DST'[0..29)] [DST'[29..44), pad, 1144 bits]
//   b_i = H(b_0 ^ b_(i-1) | i | DST')             = [32 bytes, i, DST'[0..31)] [DST'[31..44), pad, 616 bits]
// The state after the all-zero block is a constant, and the LAST block of every hash is a constant block: its message schedule
// is folded into the round constants at compile time (what hash64 does for its padding block, sha256.h KW2).  18 compressions,
// 9 of them without schedule work, no memory.
constexpr u32 c_be32(const u8* b, int i) { return ((u32)b[i] << 24) | ((u32)b[i + 1] << 16) | ((u32)b[i + 2] << 8) | (u32)b[i + 3]; }
struct Sha256KW {
    u32 v[64];
};
// K[i] + W[i] for a constant 16-word block
constexpr Sha256KW sha256_const_schedule(const u32 (&blk)[16]) {
    Sha256KW t{};
    u32 w[64] = {};
    for (int i = 0; i < 16; i++) w[i] = blk[i];
    for (int i = 16; i < 64; i++) w[i] = c_ssig1(w[i - 2]) + w[i - 7] + c_ssig0(w[i - 15]) + w[i - 16];
    for (int i = 0; i < 64; i++) t.v[i] = SHA256_K[i] + w[i];
    return t;
}
struct Sha256State {
    u32 v[8];
};
constexpr Sha256State sha256_after_zero_block() {
    u32 s[8] = {};
    for (int i = 0; i < 8; i++) s[i] = SHA256_IV[i];
    u32 a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
    for (int i = 0; i < 64; i++) {  // W = 0 throughout
        const u32 t1 = h + (c_rotr(e, 6) ^ c_rotr(e, 11) ^ c_rotr(e, 25)) + ((e & f) ^ (~e & g)) + SHA256_K[i];
        const u32 t2 = (c_rotr(a, 2) ^ c_rotr(a, 13) ^ c_rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
    }
    Sha256State r{};
    const u32 o[8] = {a, b, c, d, e, f, g, h};
    for (int i = 0; i < 8; i++) r.v[i] = s[i] + o[i];
    return r;
}
struct XmdConsts {
    Sha256State after_zpad;
    u32 b0_tail[8];   // words 8..15 of b_0's second block: 01 00 00 DST'[0], DST'[1..29)
    u32 bi_tail[8];   // words 8..15 of b_i's first block with i = 0: 00 DST'[0..3), DST'[3..31)
    Sha256KW b0_last;  // b_0's third block: DST'[29..44), 0x80, zeros, 1144 bits
    Sha256KW bi_last;  // b_i's second block: DST'[31..44), 0x80, zeros, 616 bits
};
constexpr XmdConsts make_xmd_consts() {
    XmdConsts c{};
    c.after_zpad = sha256_after_zero_block();
    const u8* d = blsc::DST_PRIME;
    u8 t0[32] = {};
    t0[0] = 0x01;
    for (int i = 0; i < 29; i++) t0[3 + i] = d[i];
    for (int i = 0; i < 8; i++) c.b0_tail[i] = c_be32(t0, 4 * i);
    u8 t1[32] = {};
    for (int i = 0; i < 31; i++) t1[1 + i] = d[i];
    for (int i = 0; i < 8; i++) c.bi_tail[i] = c_be32(t1, 4 * i);
    u8 l0[64] = {};
    for (int i = 0; i < 15; i++) l0[i] = d[29 + i];
    l0[15] = 0x80;
    u32 w0[16] = {};
    for (int i = 0; i < 16; i++) w0[i] = c_be32(l0, 4 * i);
    w0[15] = 8 * (64 + 32 + 3 + 44);
    c.b0_last = sha256_const_schedule(w0);
    u8 l1[64] = {};
    for (int i = 0; i < 13; i++) l1[i] = d[31 + i];
    l1[13] = 0x80;
    u32 w1[16] = {};
    for (int i = 0; i < 16; i++) w1[i] = c_be32(l1, 4 * i);
    w1[15] = 8 * (32 + 1 + 44);
    c.bi_last = sha256_const_schedule(w1);
    return c;
}
ECG_CONST XmdConsts XMD32 = make_xmd_consts();
// one compression whose message schedule is a compile-time constant (kw = K + W)
ECG_HD void sha256_compress_const(u32 st[8], const u32 (&kw)[64]) {
    u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; i += 8) {
        ECG_SHA_ROUND(a, b, c, d, e, f, g, h, kw[i + 0]);
        ECG_SHA_ROUND(h, a, b, c, d, e, f, g, kw[i + 1]);
        ECG_SHA_ROUND(g, h, a, b, c, d, e, f, kw[i + 2]);
        ECG_SHA_ROUND(f, g, h, a, b, c, d, e, kw[i + 3]);
        ECG_SHA_ROUND(e, f, g, h, a, b, c, d, kw[i + 4]);
        ECG_SHA_ROUND(d, e, f, g, h, a, b, c, kw[i + 5]);
        ECG_SHA_ROUND(c, d, e, f, g, h, a, b, kw[i + 6]);
        ECG_SHA_ROUND(b, c, d, e, f, g, h, a, kw[i + 7]);
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
// out: the 256 bytes of expand_message_xmd as 64 big-endian words; msg: 32 bytes (global or private memory)
ECG_HD_NOINLINE void xmd_expand_256_msg32(u32* out, const u8* msg) {
    u32 b0[8], w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        b0[i] = XMD32.after_zpad.v[i];
        w[i] = ((u32)msg[4 * i] << 24) | ((u32)msg[4 * i + 1] << 16) | ((u32)msg[4 * i + 2] << 8) | (u32)msg[4 * i + 3];
        w[8 + i] = XMD32.b0_tail[i];
    }
    sha256_block(b0, w);
    sha256_compress_const(b0, XMD32.b0_last.v);
    u32 bi[8];
#pragma unroll
    for (int i = 0; i < 8; i++) bi[i] = 0;
    for (u32 k = 1; k <= 8; k++) {
        u32 st[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            st[i] = SHA256_IV[i];
            w[i] = b0[i] ^ bi[i];  // b_1 = H(b_0 | 1 | DST'): bi = 0 on the first round
            w[8 + i] = XMD32.bi_tail[i];
        }
        w[8] |= k << 24;
        sha256_block(st, w);
        sha256_compress_const(st, XMD32.bi_last.v);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            bi[i] = st[i];
            out[8 * (k - 1) + i] = st[i];
        }
    }
}
// OS2IP of 64 bytes given as 16 big-endian words, mod p -> Montgomery (fp_from_be64)
ECG_HD Fp fp_from_be_words16(const u32* w) {
    u32 hw[12], lw[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        hw[i] = i < 4 ? w[3 - i] : 0u;
        lw[i] = w[15 - i];
    }
    return fp_add(fp_mul(raw_from_words(lw), blsc::R2), fp_mul(raw_from_words(hw), blsc::R2_384));
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

// hash_to_field: the two field elements u0, u1 of a message
ECG_HD void hash_to_field2(Fp2& u0, Fp2& u1, const u8* msg, size_t msg_len) {
    if (msg_len == 32) {  // every signing root: the register-resident form
        u32 xw[64];
        xmd_expand_256_msg32(xw, msg);
        u0 = Fp2{fp_from_be_words16(xw), fp_from_be_words16(xw + 16)};
        u1 = Fp2{fp_from_be_words16(xw + 32), fp_from_be_words16(xw + 48)};
        return;
    }
    u8 xm[256];
    xmd_expand_256(xm, msg, msg_len);
    u0 = Fp2{fp_from_be64(xm), fp_from_be64(xm + 64)};
    u1 = Fp2{fp_from_be64(xm + 128), fp_from_be64(xm + 192)};
}

ECG_HD_NOINLINE void hash_to_g2(A2& r, const u8* msg, size_t msg_len) {
    Fp2 u0, u1;
    hash_to_field2(u0, u1, msg, msg_len);
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
    Fp2 u0, u1;
    hash_to_field2(u0, u1, msg, msg_len);
    const Fp2 u = j ? u1 : u0;
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
