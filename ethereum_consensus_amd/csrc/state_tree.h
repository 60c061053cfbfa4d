// Device-resident Merkle trees of a resident BeaconState's big fields, re-hashed along DIRTY PATHS only (SURVEY.md 8f rank 2:
// "incremental (dirty-subtree) state Merkleization"; the reference re-hashes the whole state in every process_slot,
// /root/reference/ethereum-consensus/src/phase0/slot_processing.rs:67, phase0/state_transition.rs:60, although a block touches
// 10^3 .. 10^5 of a mainnet state's 10^7 leaves).
//
// Per cached field (every big field of the state plan with at least two level-0 entries: the registry, balances, the two
// participation lists, inactivity scores, the root vectors, randao mixes, slashings, the sync committees' keys, ...):
//   level 0   the hash_tree_root of every element for record kinds (lvl0: 32 B each; validators: 8 hash64 per record), or the
//             32-byte chunks of the encoding itself for packed kinds (nothing stored);
//   levels 1 .. T   every interior node, heap layout (level k at node offset 2^H - 2^(H-k+1)), T = H - 9: the level with at
//             most 512 nodes, where the fused tail's finishing job (merkle.hip run_tree_job: LDS levels, zero ladder, length
//             mix-in) takes over -- those last <= 511 + ladder hash64 are a dependent chain whatever is cached;
//   flag0     one bit per level-0 entry: marked dirty since the last root;
//   REGIONS   the subtree under each level-T node (2^T entries) is a region with a dirty list of its own (rlist / rcount).
// MARK (at patch time, one thread per touched entry): set the entry's bit; the first to set it appends the entry to its
// region's list, and the first entry of a region appends the region to the ACTIVE list.
// CLIMB (at root time, ONE WORKGROUP PER ACTIVE REGION): counters for the region's 2^T - 1 interior nodes live in LDS.  Pass 1:
// every dirty entry walks up adding 1 to each ancestor's counter and stops at the first ancestor already marked -- afterwards a
// dirty node's counter holds the number of its dirty children.  Pass 2: every dirty entry recomputes its level-0 node and walks
// up: where the parent has one dirty child it just hashes with the (clean, prefetched) sibling; where it has two it subtracts 1
// from the parent's counter: whoever takes it from 2 to 1 stops (the sibling's subtree is still on its way and will carry
// on), whoever takes it to 0 loads the sibling's fresh node, hashes and continues.  Nobody waits, and exactly the nodes on dirty
// paths are re-hashed: 4 096 dirty balances of 2^20 cost ~ 4 096 x 6 + 4 096 hash64 instead of 2^18.
// A region is one workgroup on purpose: tickets and node hand-over stay inside a CU (LDS atomics, workgroup-scope fences).  The
// first version of this file climbed with one thread per dirty entry anywhere on the chip and device-scope tickets: every level
// then costs an L2 write-back and an invalidate across the 8 XCDs -- 16 us per level against the 5 us of the hash64 itself
// (profiles/r05b_resident_probe.txt: 143 us for 9 levels).
// The same routines are compiled by g++ into tests/hostsim (sequential threads in seeded order) -- test_hostsim_merkle.py.
#pragma once
#include "merkle.h"
#include "state_plan.h"

namespace ecg {

constexpr u32 TREE_TOP_LOG = 9;     // the finishing job takes <= 2^9 nodes (TREEJOB_MAX_NODES)
constexpr u32 TREE_MAX_FIELDS = 20;
constexpr u64 TREE_MIN_ENTRIES = 2;
constexpr u32 TREE_MAX_T = 13;      // a region's counters are 4 x 2^T bytes of LDS: fields up to 2^22 entries are cached
constexpr u32 TREE_ACTIVE_CAP = TREE_MAX_FIELDS << TREE_TOP_LOG;  // every region of every field

struct TreeGeom {
    const u8* src;  // the field's bytes in the encoding (moves when an earlier list changes length)
    u64 bytes;
    u64 n0;         // level-0 entries: elements (record kinds) or 32-byte chunks (LEAF_CHUNKS)
    u8* lvl0;       // element roots, 2^H x 32 B (record kinds); null for LEAF_CHUNKS
    u8* nodes;      // levels 1 .. H, heap layout, 2^H x 32 B
    u32* flag0;     // 2^H bits
    u32* rcount;    // 2^(H-T) regions: dirty entries listed
    uint16_t* rlist;  // region r: entries rlist[r << T .. ), each the entry's index inside the region
    u32 kind, H, T;
    u32 skip;       // 1: the field is rebuilt from scratch before this climb; its stale dirty-list entries are ignored
};
struct TreeTable {
    TreeGeom f[TREE_MAX_FIELDS];
};

ECG_HD u64 tree_heap_off(u32 H, u32 k) { return (1ull << H) - (2ull << (H - k)); }  // first node of level k >= 1
ECG_HD u64 tree_level_count(u64 n0, u32 k) { return (n0 + ((1ull << k) - 1)) >> k; }
inline u32 tree_top_level(u32 kind, u32 H) {
    const u32 min_t = kind == LEAF_CHUNKS ? 1u : 0u;  // a packed field's level 0 is unaligned encoding bytes: the job starts above it
    return H > TREE_TOP_LOG + min_t ? H - TREE_TOP_LOG : min_t;
}
ECG_HD u32 tree_leaf_hashes(u32 kind) {
    switch (kind) {
        case LEAF_VALIDATORS: return 8;
        case LEAF_ETH1DATA:
        case LEAF_U64X3: return 3;
        case LEAF_CHUNKS: return 0;
        default: return 1;
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
ECG_D u32 tree_atomic_add(u32* p, u32 v) { return atomicAdd(p, v); }
ECG_D u32 tree_atomic_sub(u32* p, u32 v) { return atomicSub(p, v); }
ECG_D u32 tree_atomic_or(u32* p, u32 v) { return atomicOr(p, v); }
ECG_D u32 tree_atomic_and(u32* p, u32 v) { return atomicAnd(p, v); }
ECG_D void tree_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
#else
inline u32 tree_atomic_add(u32* p, u32 v) { const u32 o = *p; *p = o + v; return o; }
inline u32 tree_atomic_sub(u32* p, u32 v) { const u32 o = *p; *p = o - v; return o; }
inline u32 tree_atomic_or(u32* p, u32 v) { const u32 o = *p; *p = o | v; return o; }
inline u32 tree_atomic_and(u32* p, u32 v) { const u32 o = *p; *p = o & v; return o; }
inline void tree_fence() {}
#endif

// level-0 node `e` computed from the encoding
ECG_HD Node tree_leaf(const TreeGeom& g, u64 e) {
    switch (g.kind) {
        case LEAF_VALIDATORS: return ValidatorLeaves{g.src, g.bytes}(e);
        case LEAF_BYTES48: return Bytes48Leaves{g.src, g.bytes}(e);
        case LEAF_PAIR64: return Pair64Leaves{g.src, g.bytes}(e);
        case LEAF_ETH1DATA: return Eth1DataLeaves{g.src, g.bytes}(e);
        case LEAF_U64X2: return U64x2Leaves{g.src, g.bytes}(e);
        case LEAF_U64X3: return U64x3Leaves{g.src, g.bytes}(e);
        default: return ChunkLeaves{g.src, g.bytes}(e);
    }
}
// stored node i of level k (virtual nodes beyond the level's count are ladder entries)
ECG_HD Node tree_node(const TreeGeom& g, u32 k, u64 i, const ZeroTable* zt) {
    if (i >= tree_level_count(g.n0, k)) return zt->z[k];
    Node n;
    if (k == 0) {
        if (!g.lvl0) return ChunkLeaves{g.src, g.bytes}(i);
        node_load(n, g.lvl0 + 32ull * i);
        return n;
    }
    node_load(n, g.nodes + 32ull * (tree_heap_off(g.H, k) + i));
    return n;
}

constexpr u32 TREE_SLOT_SHIFT = 56;
constexpr u64 TREE_ENTRY_MASK = (1ull << TREE_SLOT_SHIFT) - 1;
ECG_HD u64 tree_local_off(u32 T, u32 k) { return (1ull << T) - (2ull << (T - k)); }  // a region's counter of level k >= 1, node 0

// MARK: one thread per (field slot, entry) the host derived from a patch; duplicates welcome
ECG_HD void tree_mark(const TreeGeom& g, u32 slot, u64 e, u32* active, u32* active_count) {
    const u32 bit = 1u << (e & 31);
    if (tree_atomic_or(&g.flag0[e >> 5], bit) & bit) return;
    const u64 r = e >> g.T;
    const u32 at = tree_atomic_add(&g.rcount[r], 1u);  // (< 2^T: an entry is listed once between two roots)
    g.rlist[(r << g.T) + at] = (uint16_t)(e & ((1ull << g.T) - 1));
    if (at == 0) {
        const u32 a = tree_atomic_add(active_count, 1u);
        if (a < TREE_ACTIVE_CAP) active[a] = (slot << 16) | (u32)r;  // (a region is listed once: never more than all of them)
    }
}

// CLIMB pass 1, per dirty entry of region: count the dirty children of every ancestor inside the region (lcnt: 2^T zeroed
// counters in LDS)
ECG_HD void tree_region_count(const TreeGeom& g, u32* lcnt, u32 le) {
    u32 i = le;
    for (u32 k = 1; k <= g.T; k++) {
        i >>= 1;
        if (tree_atomic_add(&lcnt[tree_local_off(g.T, k) + i], 1u) != 0) return;  // counted from here on up already
    }
}
// Between the passes (behind the barrier that ends pass 1, in front of the one that starts pass 2), per dirty entry: which of its
// ancestors have TWO dirty children (bit k - 1 for level k) -- read while the counters are still what pass 1 left, because pass 2
// takes them down (a counter seen at 1 later on may be a 2 whose other child has already passed).  (Loading the clean siblings
// of the whole path here as well, so that their latency runs underneath the hashes, was tried: thirteen nodes indexed by a
// runtime level live in the private segment, and the climb got slower -- profiles/r05u_resident_probe.txt.)
struct TreePath {
    u32 both;
};
ECG_HD void tree_region_path(const TreeGeom& g, const u32* lcnt, u64 region, u32 le, const ZeroTable* zt, TreePath& P) {
    const u64 e = (region << g.T) + le;
    (void)e;
    (void)zt;
    P.both = 0;
    for (u32 k = 1; k <= g.T; k++)
        if (lcnt[tree_local_off(g.T, k) + (le >> k)] == 2) P.both |= 1u << (k - 1);
}
// CLIMB pass 2, per dirty entry: returns the hash64 it performed.  Below the level where a region's paths meet an ancestor
// has ONE dirty child: nobody to wait for, nothing to hand over -- no ticket, no fences.
ECG_HD u32 tree_region_climb(const TreeGeom& g, u32* lcnt, u64 region, u32 le, const ZeroTable* zt, const TreePath& P) {
    const u64 e = (region << g.T) + le;
    tree_atomic_and(&g.flag0[e >> 5], ~(1u << (e & 31)));
    u32 hashes = tree_leaf_hashes(g.kind);
    Node x = tree_leaf(g, e);
    if (g.lvl0) node_store(x, g.lvl0 + 32ull * e);
    u64 i = e;
    u32 li = le;
    for (u32 k = 1; k <= g.T; k++) {
        const u64 p = i >> 1;
        li >>= 1;
        if ((P.both >> (k - 1)) & 1) {  // both children dirty: the last to arrive carries on, with the other's fresh node
            tree_fence();  // release (workgroup): the node stored so far is visible to the region's other lanes before the ticket is given up
            const u32 before = tree_atomic_sub(&lcnt[tree_local_off(g.T, k) + li], 1u);
            if (before != 1) return hashes;  // 2: the sibling's subtree is still on its way and carries on from here
            tree_fence();  // acquire: the sibling's node
        } else {
            lcnt[tree_local_off(g.T, k) + li] = 0;  // (nobody else touches a counter of 1; left tidy: the host simulator checks that every ticket was taken)
        }
        const Node sib = tree_node(g, k - 1, i ^ 1, zt);
        // ONE call for the wave: operands selected per lane.  (`odd ? hash64(sib, x) : hash64(x, sib)` is two calls under
        // complementary masks -- a wave holding a left and a right child ran every level twice: profiles/r05g_*)
        const bool odd = (i & 1) != 0;
        Node l, r;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            l.w[q] = odd ? sib.w[q] : x.w[q];
            r.w[q] = odd ? x.w[q] : sib.w[q];
        }
        x = hash64(l, r);
        hashes++;
        node_store(x, g.nodes + 32ull * (tree_heap_off(g.H, k) + p));
        i = p;
    }
    return hashes;
}

// REBUILD: node i of level k + D from level k, every intermediate node stored (one lane per node of level k + D)
template <int D>
struct TreeSpan {
    static ECG_HD Node run(const TreeGeom& g, u32 k, u64 i, const ZeroTable* zt) {
        if (i >= tree_level_count(g.n0, k + D)) return zt->z[k + D];
        const Node l = TreeSpan<D - 1>::run(g, k, 2 * i, zt);
        const Node r = TreeSpan<D - 1>::run(g, k, 2 * i + 1, zt);
        const Node x = hash64(l, r);
        node_store(x, g.nodes + 32ull * (tree_heap_off(g.H, k + D) + i));
        return x;
    }
};
template <>
struct TreeSpan<0> {
    static ECG_HD Node run(const TreeGeom& g, u32 k, u64 i, const ZeroTable* zt) { return tree_node(g, k, i, zt); }
};

// hash64 a full rebuild of levels 0 .. T performs
inline u64 tree_rebuild_hashes(const TreeGeom& g) {
    u64 h = g.n0 * tree_leaf_hashes(g.kind);
    for (u32 k = 1; k <= g.T; k++) h += tree_level_count(g.n0, k);
    return h;
}

}  // namespace ecg
