/* CPU restatement of the SHA-256 / SSZ Merkleization path -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * It restates, in plain C, what the reference gets from `ssz_rs` @84ef2b7 + `sha2` 0.10.8
 * (/root/reference/Cargo.toml:20,23; neither vendored): SURVEY.md Appendix A `merkleize` /
 * `mix_in_length`, and the Validator container of
 * /root/reference/ethereum-consensus/src/phase0/validator.rs:10-26.  SHA-256 compress uses the
 * x86 SHA extensions when the CPU has them (as `sha2` does), otherwise portable C.
 * Pinned against oracle/ssz.py (itself pinned to the reference fixtures) in tests/test_oracle_c.py.
 */
#include <cpuid.h>
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const uint32_t K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static const uint32_t IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                               0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))

static void compress_portable(uint32_t st[8], const uint8_t blk[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = h + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
        uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

__attribute__((target("sha,sse4.1,ssse3"))) static void compress_shani(uint32_t st[8], const uint8_t blk[64]) {
    __m128i STATE0, STATE1, MSG, TMP, MSG0, MSG1, MSG2, MSG3, ABEF_SAVE, CDGH_SAVE;
    const __m128i MASK = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    TMP = _mm_loadu_si128((const __m128i*)&st[0]);
    STATE1 = _mm_loadu_si128((const __m128i*)&st[4]);
    TMP = _mm_shuffle_epi32(TMP, 0xB1);
    STATE1 = _mm_shuffle_epi32(STATE1, 0x1B);
    STATE0 = _mm_alignr_epi8(TMP, STATE1, 8);
    STATE1 = _mm_blend_epi16(STATE1, TMP, 0xF0);
    ABEF_SAVE = STATE0;
    CDGH_SAVE = STATE1;
#define RND4(M, k)                                                          \
    MSG = _mm_add_epi32(M, _mm_loadu_si128((const __m128i*)&K[k]));         \
    STATE1 = _mm_sha256rnds2_epu32(STATE1, STATE0, MSG);                    \
    MSG = _mm_shuffle_epi32(MSG, 0x0E);                                     \
    STATE0 = _mm_sha256rnds2_epu32(STATE0, STATE1, MSG);
    MSG0 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(blk + 0)), MASK);
    MSG1 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(blk + 16)), MASK);
    MSG2 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(blk + 32)), MASK);
    MSG3 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(blk + 48)), MASK);
    RND4(MSG0, 0);
    RND4(MSG1, 4);
    RND4(MSG2, 8);
    RND4(MSG3, 12);
    for (int k = 16; k < 64; k += 16) {
#define SCHED(A, B, C, D)                                                   \
    A = _mm_sha256msg1_epu32(A, B);                                         \
    A = _mm_add_epi32(A, _mm_alignr_epi8(D, C, 4));                         \
    A = _mm_sha256msg2_epu32(A, D);
        SCHED(MSG0, MSG1, MSG2, MSG3);
        RND4(MSG0, k);
        SCHED(MSG1, MSG2, MSG3, MSG0);
        RND4(MSG1, k + 4);
        SCHED(MSG2, MSG3, MSG0, MSG1);
        RND4(MSG2, k + 8);
        SCHED(MSG3, MSG0, MSG1, MSG2);
        RND4(MSG3, k + 12);
    }
    STATE0 = _mm_add_epi32(STATE0, ABEF_SAVE);
    STATE1 = _mm_add_epi32(STATE1, CDGH_SAVE);
    TMP = _mm_shuffle_epi32(STATE0, 0x1B);
    STATE1 = _mm_shuffle_epi32(STATE1, 0xB1);
    STATE0 = _mm_blend_epi16(TMP, STATE1, 0xF0);
    STATE1 = _mm_alignr_epi8(STATE1, TMP, 8);
    _mm_storeu_si128((__m128i*)&st[0], STATE0);
    _mm_storeu_si128((__m128i*)&st[4], STATE1);
}

static int g_have_shani = -1;
int oc_have_shani(void) {
    if (g_have_shani < 0) {
        unsigned a, b, c, d;
        g_have_shani = 0;
        if (__get_cpuid_count(7, 0, &a, &b, &c, &d)) g_have_shani = (b >> 29) & 1;
    }
    return g_have_shani;
}
int oc_force_portable(int on) {
    if (on) g_have_shani = 0; else { g_have_shani = -1; oc_have_shani(); }
    return g_have_shani;
}

static void compress(uint32_t st[8], const uint8_t blk[64]) {
    if (oc_have_shani()) compress_shani(st, blk); else compress_portable(st, blk);
}

void oc_sha256(const uint8_t* data, uint64_t len, uint8_t out[32]) {
    uint32_t st[8];
    memcpy(st, IV, sizeof(st));
    uint64_t off = 0;
    for (; off + 64 <= len; off += 64) compress(st, data + off);
    uint8_t tail[128];
    uint64_t r = len - off;
    memset(tail, 0, sizeof(tail));
    memcpy(tail, data + off, r);
    tail[r] = 0x80;
    uint64_t tl = (r + 9 <= 64) ? 64 : 128;
    uint64_t bits = len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    compress(st, tail);
    if (tl == 128) compress(st, tail + 64);
    for (int i = 0; i < 8; i++) { out[4*i] = st[i] >> 24; out[4*i+1] = st[i] >> 16; out[4*i+2] = st[i] >> 8; out[4*i+3] = st[i]; }
}

static void hash64(const uint8_t* l, const uint8_t* r, uint8_t out[32]) {
    static const uint8_t PAD[64] = {0x80, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 0};
    uint8_t blk[64];
    uint32_t st[8];
    memcpy(blk, l, 32);
    memcpy(blk + 32, r, 32);
    memcpy(st, IV, sizeof(st));
    compress(st, blk);
    compress(st, PAD);
    for (int i = 0; i < 8; i++) { out[4*i] = st[i] >> 24; out[4*i+1] = st[i] >> 16; out[4*i+2] = st[i] >> 8; out[4*i+3] = st[i]; }
}

static uint8_t ZH[65][32];
static int zh_ready = 0;
static void zh_init(void) {
    if (zh_ready) return;
    memset(ZH[0], 0, 32);
    for (int d = 1; d <= 64; d++) hash64(ZH[d - 1], ZH[d - 1], ZH[d]);
    zh_ready = 1;
}

static unsigned depth_for(uint64_t limit) {
    unsigned d = 0;
    while (d < 64 && (1ull << d) < limit) d++;
    return d;
}

/* Appendix A merkleize over n 32-byte chunks (in place on a scratch copy), limit -> depth */
uint64_t oc_merkleize_chunks(const uint8_t* chunks, uint64_t n, uint64_t limit, uint8_t root[32]) {
    zh_init();
    unsigned depth = depth_for(limit ? limit : n);
    uint64_t hashes = 0;
    if (n == 0) { memcpy(root, ZH[depth], 32); return 0; }
    uint8_t* buf = (uint8_t*)malloc(32 * (n + 1));
    memcpy(buf, chunks, 32 * n);
    uint64_t m = n;
    for (unsigned d = 0; d < depth; d++) {
        if (m & 1) { memcpy(buf + 32 * m, ZH[d], 32); m++; }
        for (uint64_t i = 0; i < m; i += 2) hash64(buf + 32 * i, buf + 32 * (i + 1), buf + 16 * i);
        m >>= 1;
        hashes += m;
    }
    memcpy(root, buf, 32);
    free(buf);
    return hashes;
}

void oc_mix_in_length(uint8_t root[32], uint64_t len) {
    uint8_t l[32];
    memset(l, 0, 32);
    for (int i = 0; i < 8; i++) l[i] = (uint8_t)(len >> (8 * i));
    hash64(root, l, root);
}

uint64_t oc_merkleize_bytes(const uint8_t* data, uint64_t n_bytes, uint64_t limit_chunks, int mix, uint64_t len, uint8_t root[32]) {
    uint64_t n = (n_bytes + 31) / 32;
    uint8_t* buf = (uint8_t*)calloc(n ? n : 1, 32);
    memcpy(buf, data, n_bytes);
    uint64_t h = oc_merkleize_chunks(buf, n, limit_chunks ? limit_chunks : n, root);
    free(buf);
    if (mix) { oc_mix_in_length(root, len); h++; }
    return h;
}

/* hash_tree_root(Validator) from the 121-byte record (phase0/validator.rs:10-26) */
void oc_htr_validator(const uint8_t r[121], uint8_t root[32]) {
    uint8_t leaf[8][32], t[4][32], pk2[32];
    memset(leaf, 0, sizeof(leaf));
    memset(pk2, 0, 32);
    memcpy(pk2, r + 32, 16);
    hash64(r, pk2, leaf[0]);
    memcpy(leaf[1], r + 48, 32);
    memcpy(leaf[2], r + 80, 8);
    leaf[3][0] = r[88];
    for (int k = 0; k < 4; k++) memcpy(leaf[4 + k], r + 89 + 8 * k, 8);
    for (int k = 0; k < 4; k++) hash64(leaf[2 * k], leaf[2 * k + 1], t[k]);
    hash64(t[0], t[1], t[0]);
    hash64(t[2], t[3], t[2]);
    hash64(t[0], t[2], root);
}

/* hash_tree_root(List<Validator, limit>) -> number of hash64 performed */
uint64_t oc_htr_validators(const uint8_t* ssz121, uint64_t n, uint64_t limit, uint8_t root[32]) {
    uint8_t* roots = (uint8_t*)malloc(32 * (n ? n : 1));
    for (uint64_t i = 0; i < n; i++) oc_htr_validator(ssz121 + 121 * i, roots + 32 * i);
    uint64_t h = 8 * n + oc_merkleize_chunks(roots, n, limit, root);
    free(roots);
    oc_mix_in_length(root, n);
    return h + 1;
}

/* root of the aligned subtree of `width` (a power of two) validators holding the first n <= width records: no length mix-in.
   One thread's share when bench.py times this restatement on several host threads ("1 and 8 threads"). */
uint64_t oc_validators_subtree_root(const uint8_t* ssz121, uint64_t n, uint64_t width, uint8_t root[32]) {
    uint8_t* roots = (uint8_t*)malloc(32 * (n ? n : 1));
    for (uint64_t i = 0; i < n; i++) oc_htr_validator(ssz121 + 121 * i, roots + 32 * i);
    uint64_t h = 8 * n + oc_merkleize_chunks(roots, n, width, root);
    free(roots);
    return h;
}
