// SSZ Merkleization lane programs (replaces ssz_rs `merkleize` / `mix_in_length` under every
// `.hash_tree_root()` of the reference, e.g. /root/reference/ethereum-consensus/src/phase0/
// slot_processing.rs:67,75; algorithm = SURVEY.md Appendix A).
//
// Layout: a wave covers a tile of 64 * 2^D consecutive chunks; every LANE owns the 2^D
// consecutive chunks of one aligned subtree and reduces them depth-first entirely in
// registers (2^D - 1 hash64, no LDS, no cross-lane traffic, all 64 lanes busy), then stores
// one 32-byte node.  A lane streams its own contiguous 32*2^D bytes, so every fetched cache
// line is consumed exactly once (leaf buffer read once from HBM); the 64 node stores of a wave
// are contiguous (2 KiB).  Interior nodes below level D never touch memory.
#pragma once
#include "sha256.h"

namespace ecg {

// zero-subtree ladder Z_0..Z_64, device-resident, filled at init by a
// GPU kernel (no host hashing).
struct ZeroTable {
    Node z[65];
};

ECG_HD u32 funnel_bytes(u32 lo, u32 hi, u32 byte_shift) {
    // bytes [byte_shift, byte_shift+4) of the little-endian 8-byte value hi:lo -> v_alignbyte_b32
    u64 v = ((u64)hi << 32) | lo;
    return (u32)(v >> (8 * byte_shift));
}

// Load `nw` little-endian dwords that start at byte `off` of the buffer [base, base+total);
// bytes at or beyond `total` read as zero.  Only naturally aligned dwords that contain at
// least one in-range byte are touched, so unaligned SSZ slices of a larger buffer are safe.
template <int NW>
ECG_HD void load_bytes_le(u32 (&out)[NW], const u8* base, u64 off, u64 total) {
    const u8* addr = base + off;
    const u64 a = (u64)addr;
    const u32 sh = (u32)(a & 3);
    const u8* end = base + total;
    if (sh == 0 && off + 4ull * NW <= total) {
        const u32* q = reinterpret_cast<const u32*>(addr);
#pragma unroll
        for (int i = 0; i < NW; i++) out[i] = q[i];
        return;
    }
    const u32* q = reinterpret_cast<const u32*>(a - sh);
    u32 raw[NW + 1];
#pragma unroll
    for (int i = 0; i < NW + 1; i++) {
        const u8* p = reinterpret_cast<const u8*>(q + i);
        raw[i] = (p < end && off < total) ? q[i] : 0u;
    }
    const u64 rem = off < total ? total - off : 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        u32 v = funnel_bytes(raw[i], raw[i + 1], sh);
        u64 valid = rem > 4ull * i ? rem - 4ull * i : 0;
        if (valid < 4) v &= valid == 0 ? 0u : (0xffffffffu >> (8 * (4 - (u32)valid)));
        out[i] = v;
    }
}

// ---- leaf functors: produce the level-0 node with index `idx` ------------------------------

// packed chunks: node idx = bytes [32 idx, 32 idx + 32) of the buffer, zero padded (`pack`)
struct ChunkLeaves {
    const u8* base;
    u64 total_bytes;
    ECG_HD Node operator()(u64 idx) const {
        u32 d[8];
        load_bytes_le<8>(d, base, idx * 32, total_bytes);
        Node n;
#pragma unroll
        for (int i = 0; i < 8; i++) n.w[i] = ecg_bswap32(d[i]);
        return n;
    }
};

// nodes already produced by a previous pass (32-byte aligned workspace)
struct NodeLeaves {
    const u8* base;
    ECG_HD Node operator()(u64 idx) const {
        Node n;
        node_load(n, base + idx * 32);
        return n;
    }
};

// hash_tree_root(Validator) from the 121-byte SSZ record
// (/root/reference/ethereum-consensus/src/phase0/validator.rs:10-26): leaves
//   htr(pubkey 48 B) | withdrawal_credentials | effective_balance | slashed |
//   activation_eligibility_epoch | activation_epoch | exit_epoch | withdrawable_epoch
// = 1 + 7 hash64 per validator.
// The validator's root from its record, read through `word(i)` = little-endian dword i of the record (i < 31; the last three
// bytes of dword 30 belong to the next record and are not used).  Words are asked for right before the hash64 that consumes
// them, and `word.after(node)` is told each intermediate result: a source that fetches on demand (the LDS-staged registry
// pass) ties its next fetches to that result, so that no copy of the record sits in registers across the calls.
template <class Words>
ECG_HD Node validator_root_from_words(Words& word) {
    Node a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) a.w[i] = ecg_bswap32(word(i));          // pubkey[0..32)
#pragma unroll
    for (int i = 0; i < 4; i++) { b.w[i] = ecg_bswap32(word(8 + i)); b.w[4 + i] = 0; }  // pubkey[32..48) || 0^16
    a = hash64(a, b);
    word.after(a);
#pragma unroll
    for (int i = 0; i < 8; i++) b.w[i] = ecg_bswap32(word(12 + i));   // withdrawal_credentials
    Node left = hash64(a, b);
    word.after(left);
    node_zero(a);
    node_zero(b);
    a.w[0] = ecg_bswap32(word(20));                                   // effective_balance u64 LE
    a.w[1] = ecg_bswap32(word(21));
    b.w[0] = ecg_bswap32(word(22) & 0xffu);                           // slashed: byte 88
    left = hash64(left, hash64(a, b));
    word.after(left);
    // four u64 epochs at bytes 89,97,105,113: one byte past a dword boundary
    Node right;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        node_zero(a);
        node_zero(b);
        const u32 w0 = word(22 + 4 * k), w1 = word(23 + 4 * k), w2 = word(24 + 4 * k), w3 = word(25 + 4 * k), w4 = word(26 + 4 * k);
        a.w[0] = ecg_bswap32(funnel_bytes(w0, w1, 1));
        a.w[1] = ecg_bswap32(funnel_bytes(w1, w2, 1));
        b.w[0] = ecg_bswap32(funnel_bytes(w2, w3, 1));
        b.w[1] = ecg_bswap32(funnel_bytes(w3, w4, 1));
        const Node h = hash64(a, b);
        word.after(h);
        right = k == 0 ? h : hash64(right, h);
    }
    return hash64(left, right);
}

struct WordArray31 {
    u32 d[31];
    ECG_HD u32 operator()(int i) const { return d[i]; }
    ECG_HD void after(const Node&) {}
};

struct ValidatorLeaves {
    const u8* base;  // n * 121 bytes
    u64 total_bytes;
    ECG_HD Node operator()(u64 idx) const {
        WordArray31 w;
        load_bytes_le<31>(w.d, base, idx * 121, total_bytes);  // 124 bytes, the last 3 are the next record
        return validator_root_from_words(w);
    }
};

// ---- the registry's leaf pass staged through LDS (merkle.hip k_merkle_pass<2, ValidatorLeaves>; the host simulator runs the
// same addressing, tests/hostsim hs_staged_validator_wave) -------------------------------------------------------------------
constexpr u32 VAL_STEP_BYTES = 64 * 121;                 // a wave's step: 64 records = 484 x 16 bytes
constexpr u32 VAL_STAGE_VECS = VAL_STEP_BYTES / 16 + 1;  // + the vector the step's misalignment spills into
// a record in the stage, read a dword pair at a time where the words are used
struct StagedRecord {
    const u32* stage;  // the wave's stage
    u32 at;            // the aligned dword that holds the record's first byte
    u32 sh;            // ... and that byte's position in it
    ECG_HD u32 operator()(int i) const {
        // (dword 30 is asked for its first byte only: the dword after it, which may lie past the stage, is never needed)
        return i < 30 ? funnel_bytes(stage[at + i], stage[at + i + 1], sh) : funnel_bytes(stage[at + 30], 0u, sh);
    }
    ECG_HD void after(const Node& n) {
#if defined(__HIP_DEVICE_COMPILE__)
        // the next fetches "depend" on the hash just computed: they stay behind the call (hash64 is pure, and without this the
        // compiler reads the whole record up front and carries 31 words across seven calls)
        asm volatile("" : "+v"(at) : "v"(n.w[0]));
#else
        (void)n;
#endif
    }
};
// record `lane` of a step whose first record starts `adj` bytes into the stage
ECG_HD StagedRecord staged_record(const u32* stage, u32 adj, u32 lane) {
    const u32 o = adj + 121 * lane;
    return StagedRecord{stage, o >> 2, o & 3};
}
// the transposition of a wave's 256 record roots through the stage, four words at a time: word w of record v
ECG_HD u32 staged_root_dword(u32 w, u32 v) { return 256 * w + v; }

// hash_tree_root(ByteVector<48>) for packed 48-byte records (BlsPublicKey vectors of
// SyncCommittee, /root/reference/ethereum-consensus/src/altair/sync.rs:17-22)
struct Bytes48Leaves {
    const u8* base;
    u64 total_bytes;
    ECG_HD Node operator()(u64 idx) const {
        u32 d[12];
        load_bytes_le<12>(d, base, idx * 48, total_bytes);
        Node a, b;
#pragma unroll
        for (int i = 0; i < 8; i++) a.w[i] = ecg_bswap32(d[i]);
#pragma unroll
        for (int i = 0; i < 4; i++) { b.w[i] = ecg_bswap32(d[8 + i]); b.w[4 + i] = 0; }
        return hash64(a, b);
    }
};

// two-chunk containers packed as 64-byte records (Checkpoint-like pairs of roots,
// HistoricalSummary: /root/reference/ethereum-consensus/src/phase0/beacon_state.rs:42-45)
struct Pair64Leaves {
    const u8* base;
    u64 total_bytes;
    ECG_HD Node operator()(u64 idx) const {
        u32 d[16];
        load_bytes_le<16>(d, base, idx * 64, total_bytes);
        Node a, b;
#pragma unroll
        for (int i = 0; i < 8; i++) { a.w[i] = ecg_bswap32(d[i]); b.w[i] = ecg_bswap32(d[8 + i]); }
        return hash64(a, b);
    }
};

// hash_tree_root(Eth1Data) from the 72-byte record deposit_root | deposit_count u64 | block_hash
// (/root/reference/ethereum-consensus/src/phase0/operations.rs:66-71): 3 leaves padded to 4.
struct Eth1DataLeaves {
    const u8* base;
    u64 total_bytes;
    ECG_HD Node operator()(u64 idx) const {
        u32 d[18];
        load_bytes_le<18>(d, base, idx * 72, total_bytes);
        Node a, b, c, z;
        node_zero(b);
        node_zero(z);
#pragma unroll
        for (int i = 0; i < 8; i++) { a.w[i] = ecg_bswap32(d[i]); c.w[i] = ecg_bswap32(d[10 + i]); }
        b.w[0] = ecg_bswap32(d[8]);
        b.w[1] = ecg_bswap32(d[9]);
        return hash64(hash64(a, b), hash64(c, z));
    }
};

// hash_tree_root of a container of two uint64 from its 16-byte record: hash64(chunk(a), chunk(b)) -- electra's
// PendingBalanceDeposit {index, amount} and PendingConsolidation {source_index, target_index}
// (/root/reference/ethereum-consensus/src/electra/beacon_state.rs:27-58)
struct U64x2Leaves {
    const u8* base;
    u64 total_bytes;
    ECG_HD Node operator()(u64 idx) const {
        u32 d[4];
        load_bytes_le<4>(d, base, idx * 16, total_bytes);
        Node a, b;
        node_zero(a);
        node_zero(b);
        a.w[0] = ecg_bswap32(d[0]);
        a.w[1] = ecg_bswap32(d[1]);
        b.w[0] = ecg_bswap32(d[2]);
        b.w[1] = ecg_bswap32(d[3]);
        return hash64(a, b);
    }
};
// ... and of three uint64 from its 24-byte record (3 leaves padded to 4): electra's PendingPartialWithdrawal
// {index, amount, withdrawable_epoch} (electra/beacon_state.rs:38-47)
struct U64x3Leaves {
    const u8* base;
    u64 total_bytes;
    ECG_HD Node operator()(u64 idx) const {
        u32 d[6];
        load_bytes_le<6>(d, base, idx * 24, total_bytes);
        Node a, b, c, z;
        node_zero(a);
        node_zero(b);
        node_zero(c);
        node_zero(z);
        a.w[0] = ecg_bswap32(d[0]);
        a.w[1] = ecg_bswap32(d[1]);
        b.w[0] = ecg_bswap32(d[2]);
        b.w[1] = ecg_bswap32(d[3]);
        c.w[0] = ecg_bswap32(d[4]);
        c.w[1] = ecg_bswap32(d[5]);
        return hash64(hash64(a, b), hash64(c, z));
    }
};

// ---- in-lane depth-first subtree ------------------------------------------------------------
// Root of the aligned subtree of height K whose first level-0 node is `first`; level-0 nodes
// with index >= n are virtual: an entirely virtual subtree of height k at absolute level
// `level0 + k` is the ladder entry Z[level0 + k] (odd tails pair with Z_d, Appendix A).
template <int K, class Leaf>
struct Subtree {
    static ECG_HD Node run(const Leaf& leaf, u64 first, u64 n, const ZeroTable* zt, int level0) {
        if (first >= n) return zt->z[level0 + K];
        Node l = Subtree<K - 1, Leaf>::run(leaf, first, n, zt, level0);
        Node r = Subtree<K - 1, Leaf>::run(leaf, first + (1ull << (K - 1)), n, zt, level0);
        return hash64(l, r);
    }
};
template <class Leaf>
struct Subtree<0, Leaf> {
    static ECG_HD Node run(const Leaf& leaf, u64 first, u64 n, const ZeroTable* zt, int level0) {
        if (first >= n) return zt->z[level0];
        return leaf(first);
    }
};

// One lane of a pass: out node `gid` = subtree of height D over level-0 nodes [gid<<D, (gid+1)<<D).
template <int D, class Leaf>
ECG_HD void lane_pass(const Leaf& leaf, u64 gid, u64 n_in, u8* out, const ZeroTable* zt, int level0) {
    Node r = Subtree<D, Leaf>::run(leaf, gid << D, n_in, zt, level0);
    node_store(r, out + gid * 32);
}

ECG_HD Node len_chunk(u64 len) {
    // u256 little-endian length as a node (mix_in_length)
    Node n;
    node_zero(n);
    n.w[0] = ecg_bswap32((u32)len);
    n.w[1] = ecg_bswap32((u32)(len >> 32));
    return n;
}

// A finishing job: `n` nodes at absolute level `level` -> climb to `depth` with the zero
// ladder -> optional mix_in_length.  Executed by one workgroup (k_tree_jobs) or by the
// hostsim with the same per-lane steps.
struct TreeJob {
    u64 in_off;    // byte offset of the first input node in the job buffer (32-byte aligned)
    u64 out_off;   // byte offset of the 32-byte result
    u64 mix_len;   // length to mix in
    u32 n;         // number of input nodes (<= TREEJOB_MAX_NODES)
    u32 level;     // absolute level of the input nodes
    u32 depth;     // absolute level of the root = ceil(log2(limit))
    u32 mix;       // 1: mix_in_length(root, mix_len)
};
constexpr u32 TREEJOB_MAX_NODES = 512;

// A tile stage: workgroup `first_wg + t` reduces the 1024 level-`level0` nodes [1024 t, 1024 t + 1024) of one
// field -- 4 per lane depth-first in registers (the field's leaf functor runs here), then up to 8 levels through
// LDS -- and stores ONE node at level min(top, level0 + 10).  Several fields share one launch (k_tree_tiles),
// so a narrow tree costs two dependent launches (tiles, finishing job) instead of one per level.
struct TileDesc {
    const u8* in;    // field bytes (leaf functor input) or level-`level0` nodes
    u64 in_bytes;
    u64 n0;          // number of level-`level0` nodes
    u8* out;         // one 32-byte node per tile
    u32 kind;        // LeafKind (not LEAF_VALIDATORS: that functor has a pass of its own)
    u32 level0;      // absolute level of the inputs
    u32 top;         // absolute level of the field's root (= depth)
    u32 first_wg;    // first workgroup of this field in the launch
};
constexpr u32 TILE_LANES = 256, TILE_D = 2, TILE_NODES = TILE_LANES << TILE_D, TILE_LEVELS = TILE_D + 8;
constexpr u64 TILE_MAX_IN = (u64)TREEJOB_MAX_NODES * TILE_NODES;

}  // namespace ecg
