// Swap-or-not shuffling lane programs: the committee computation of the reference,
// /root/reference/ethereum-consensus/src/phase0/helpers.rs:249-282 (compute_shuffled_index) and :287-360
// (compute_shuffled_indices, the whole-list form used by the `optimized` feature: out[i] = in[shuffled_index(i)]).
// Both hash inputs are shorter than 56 bytes, i.e. ONE SHA-256 block each:
//     pivot_r        = LE64(SHA-256(seed || r)[0..8]) mod n                       (33 bytes)
//     source_{r,p}   = SHA-256(seed || r || LE32(p)),  p = position / 256          (37 bytes)
// The list form needs, for every round, the source blocks of ALL positions: rounds x ceil(n / 256) independent
// hashes (one lane each), after which every index walks its 90 rounds independently against that table.
#pragma once
#include "sha256.h"

namespace ecg {

// seed as 8 big-endian words
struct ShuffleSeed {
    u32 w[8];
};

ECG_HD void shuffle_hash_block(u32 out[8], const ShuffleSeed& seed, u32 round, bool with_position, u32 position) {
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = seed.w[i];
#pragma unroll
    for (int i = 8; i < 16; i++) w[i] = 0;
    if (!with_position) {
        w[8] = (round << 24) | 0x00800000u;  // byte 32 = round, byte 33 = 0x80
        w[15] = 33 * 8;
    } else {
        // bytes 32 = round, 33..36 = position little-endian, 37 = 0x80
        w[8] = (round << 24) | ((position & 0xff) << 16) | (((position >> 8) & 0xff) << 8) | ((position >> 16) & 0xff);
        w[9] = ((position >> 24) << 24) | 0x00800000u;
        w[15] = 37 * 8;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = SHA256_IV[i];
    sha256_compress(out, w);
}

ECG_HD u64 shuffle_pivot(const ShuffleSeed& seed, u32 round, u64 n) {
    u32 d[8];
    shuffle_hash_block(d, seed, round, false, 0);
    const u64 lo = ecg_bswap32(d[0]), hi = ecg_bswap32(d[1]);  // digest bytes 0..7, little-endian u64
    return ((hi << 32) | lo) % n;
}

// table[(round * n_blocks + p) * 8 + k] = word k (big-endian-interpreted) of source_{round,p}
ECG_HD u32 shuffle_source_bit(const u32* table, u64 n_blocks, u32 round, u64 position) {
    const u32* src = table + ((u64)round * n_blocks + (position >> 8)) * 8;
    const u32 byte_idx = (u32)(position & 0xff) >> 3;
    const u32 byte = (src[byte_idx >> 2] >> (8 * (3 - (byte_idx & 3)))) & 0xff;
    return (byte >> (position & 7)) & 1;
}

// compute_shuffled_index (helpers.rs:249-282) against precomputed pivots and sources
ECG_HD u64 shuffled_index(u64 index, u64 n, u32 rounds, const u64* pivots, const u32* table, u64 n_blocks) {
    for (u32 r = 0; r < rounds; r++) {
        const u64 flip = (pivots[r] + n - index) % n;
        const u64 position = index > flip ? index : flip;
        if (shuffle_source_bit(table, n_blocks, r, position)) index = flip;
    }
    return index;
}

}  // namespace ecg
