// SHA-256 for the Merkle path (replaces `sha2` under ssz_rs and crypto::hash,
// /root/reference/ethereum-consensus/src/crypto/bls.rs:12-20).
//
// A Merkle node is kept in registers as the 8 big-endian-interpreted state words; byte swaps
// happen only where a node crosses HBM.  hash64(left,right) = SHA-256 of exactly 64 bytes: the
// data block plus ONE constant padding block whose expanded message schedule is folded into the
// round constants at compile time (KW2), so the second compression has no schedule work.
//
// Op count per hash64 on gfx950 (v_alignbit / v_xor3 / v_bfi / v_add3): 2*64 rounds * 15 +
// 48 schedule words * 10 ~= 2400 VALU ops, 0 LDS, 64 B in, 32 B out.
#pragma once
#include "common.h"

namespace ecg {

struct Sha256Consts {
    u32 k[64];
    u32 kw2[64];  // K[i] + W2[i], W2 = schedule of the padding block of a 64-byte message
};

constexpr u32 SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

constexpr u32 SHA256_IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                              0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

constexpr u32 c_rotr(u32 x, int n) { return (x >> n) | (x << (32 - n)); }
constexpr u32 c_ssig0(u32 x) { return c_rotr(x, 7) ^ c_rotr(x, 18) ^ (x >> 3); }
constexpr u32 c_ssig1(u32 x) { return c_rotr(x, 17) ^ c_rotr(x, 19) ^ (x >> 10); }

struct KW2Table {
    u32 v[64];
    constexpr KW2Table() : v() {
        u32 w[64] = {};
        w[0] = 0x80000000u;
        w[15] = 512;
        for (int i = 16; i < 64; i++) w[i] = c_ssig1(w[i - 2]) + w[i - 7] + c_ssig0(w[i - 15]) + w[i - 16];
        for (int i = 0; i < 64; i++) v[i] = SHA256_K[i] + w[i];
    }
};
constexpr KW2Table SHA256_KW2 = KW2Table();

ECG_HD u32 rotr(u32 x, int n) { return (x >> n) | (x << (32 - n)); }  // -> v_alignbit_b32
ECG_HD u32 bsig0(u32 x) { return ecg_xor3(rotr(x, 2), rotr(x, 13), rotr(x, 22)); }
ECG_HD u32 bsig1(u32 x) { return ecg_xor3(rotr(x, 6), rotr(x, 11), rotr(x, 25)); }
ECG_HD u32 ssig0(u32 x) { return ecg_xor3(rotr(x, 7), rotr(x, 18), x >> 3); }
ECG_HD u32 ssig1(u32 x) { return ecg_xor3(rotr(x, 17), rotr(x, 19), x >> 10); }
ECG_HD u32 ch(u32 e, u32 f, u32 g) { return ecg_sel(e, f, g); }   // v_bitop3_b32 0xCA
ECG_HD u32 maj(u32 a, u32 b, u32 c) { return ecg_maj(a, b, c); }  // v_bitop3_b32 0xE8

#define ECG_SHA_ROUND(a, b, c, d, e, f, g, h, kw)            \
    {                                                        \
        u32 t1 = (h) + bsig1(e) + ch(e, f, g) + (kw);        \
        u32 t2 = bsig0(a) + maj(a, b, c);                    \
        (d) += t1;                                           \
        (h) = t1 + t2;                                       \
    }

// One compression with a 16-word message block (schedule computed in place, fully unrolled).
ECG_HD void sha256_compress(u32 st[8], u32 w[16]) {
    u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; i += 8) {
        if (i >= 16) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                int t = (i + j) & 15;
                w[t] += ssig1(w[(t + 14) & 15]) + w[(t + 9) & 15] + ssig0(w[(t + 1) & 15]);
            }
        }
        ECG_SHA_ROUND(a, b, c, d, e, f, g, h, SHA256_K[i + 0] + w[(i + 0) & 15]);
        ECG_SHA_ROUND(h, a, b, c, d, e, f, g, SHA256_K[i + 1] + w[(i + 1) & 15]);
        ECG_SHA_ROUND(g, h, a, b, c, d, e, f, SHA256_K[i + 2] + w[(i + 2) & 15]);
        ECG_SHA_ROUND(f, g, h, a, b, c, d, e, SHA256_K[i + 3] + w[(i + 3) & 15]);
        ECG_SHA_ROUND(e, f, g, h, a, b, c, d, SHA256_K[i + 4] + w[(i + 4) & 15]);
        ECG_SHA_ROUND(d, e, f, g, h, a, b, c, SHA256_K[i + 5] + w[(i + 5) & 15]);
        ECG_SHA_ROUND(c, d, e, f, g, h, a, b, SHA256_K[i + 6] + w[(i + 6) & 15]);
        ECG_SHA_ROUND(b, c, d, e, f, g, h, a, SHA256_K[i + 7] + w[(i + 7) & 15]);
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// Compression of the constant padding block of a 64-byte message.
ECG_HD void sha256_compress_pad64(u32 st[8]) {
    u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; i += 8) {
        ECG_SHA_ROUND(a, b, c, d, e, f, g, h, SHA256_KW2.v[i + 0]);
        ECG_SHA_ROUND(h, a, b, c, d, e, f, g, SHA256_KW2.v[i + 1]);
        ECG_SHA_ROUND(g, h, a, b, c, d, e, f, SHA256_KW2.v[i + 2]);
        ECG_SHA_ROUND(f, g, h, a, b, c, d, e, SHA256_KW2.v[i + 3]);
        ECG_SHA_ROUND(e, f, g, h, a, b, c, d, SHA256_KW2.v[i + 4]);
        ECG_SHA_ROUND(d, e, f, g, h, a, b, c, SHA256_KW2.v[i + 5]);
        ECG_SHA_ROUND(c, d, e, f, g, h, a, b, SHA256_KW2.v[i + 6]);
        ECG_SHA_ROUND(b, c, d, e, f, g, h, a, SHA256_KW2.v[i + 7]);
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// A Merkle node in registers.
struct Node {
    u32 w[8];
};

// hash64 = SHA-256(l || r).  Not inlined: one copy of the ~2400-instruction body per kernel;
// arguments and result travel in VGPRs.
ECG_HD_NOINLINE Node hash64(Node l, Node r) {
    u32 st[8], w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { st[i] = SHA256_IV[i]; w[i] = l.w[i]; w[8 + i] = r.w[i]; }
    sha256_compress(st, w);
    sha256_compress_pad64(st);
    Node out;
#pragma unroll
    for (int i = 0; i < 8; i++) out.w[i] = st[i];
    return out;
}

// node <-> 32 memory bytes
ECG_HD void node_load(Node& n, const u8* p) {
    const u32* q = reinterpret_cast<const u32*>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) n.w[i] = ecg_bswap32(q[i]);
}
ECG_HD void node_store(const Node& n, u8* p) {
    u32* q = reinterpret_cast<u32*>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = ecg_bswap32(n.w[i]);
}
ECG_HD void node_zero(Node& n) {
#pragma unroll
    for (int i = 0; i < 8; i++) n.w[i] = 0;
}

// General SHA-256 over a byte string (crypto::hash; expand_message_xmd).  Sequential.
struct Sha256Stream {
    u32 st[8];
    u32 w[16];
    u32 fill;   // bytes currently in w (0..63)
    u64 total;  // bytes absorbed
};
ECG_HD void sha256_init(Sha256Stream& s) {
    for (int i = 0; i < 8; i++) s.st[i] = SHA256_IV[i];
    for (int i = 0; i < 16; i++) s.w[i] = 0;
    s.fill = 0;
    s.total = 0;
}
ECG_HD void sha256_put(Sha256Stream& s, u8 byte) {
    u32 idx = s.fill >> 2, sh = (3 - (s.fill & 3)) * 8;
    // dynamic index into w: wave-uniform in all our uses
    for (int i = 0; i < 16; i++)
        if ((u32)i == idx) s.w[i] |= (u32)byte << sh;
    s.fill++;
    s.total++;
    if (s.fill == 64) {
        sha256_compress(s.st, s.w);
        for (int i = 0; i < 16; i++) s.w[i] = 0;
        s.fill = 0;
    }
}
ECG_HD void sha256_update(Sha256Stream& s, const u8* p, size_t n) {
    for (size_t i = 0; i < n; i++) sha256_put(s, p[i]);
}
ECG_HD void sha256_final(Sha256Stream& s, u32 digest_words[8]) {
    u64 bits = s.total * 8;
    sha256_put(s, 0x80);
    while (s.fill != 56) sha256_put(s, 0);
    for (int i = 7; i >= 0; i--) sha256_put(s, (u8)(bits >> (8 * i)));
    for (int i = 0; i < 8; i++) digest_words[i] = s.st[i];
}

}  // namespace ecg
