// tests/hostsim: CPU-side simulator of the gfx950 kernels -- TEST INFRASTRUCTURE ONLY.
//
// Compiles the very same ECG_HD lane programs the GPU executes (ethereum_consensus_amd/csrc/*.h)
// with g++ and runs them lane by lane, so that the CPU test-suite (`-m "not gpu"`) can check the
// device arithmetic against oracle/ in a container without a GPU.  Nothing in the product
// (libecgpu.so, ethereum_consensus_amd/*.py) links or loads this file.
#include <cstring>
#include <vector>

#include "merkle.h"
#include "state_plan.h"
#include "ssz_plan.h"
#include "shuffle.h"
#include "state_tree.h"

using namespace ecg;

static ZeroTable g_zt;
static bool g_zt_ready = false;
static const ZeroTable* zt() {
    if (!g_zt_ready) {
        Node z;
        node_zero(z);
        g_zt.z[0] = z;
        for (int d = 1; d <= 64; d++) {
            z = hash64(z, z);
            g_zt.z[d] = z;
        }
        g_zt_ready = true;
    }
    return &g_zt;
}

template <class Leaf>
static void run_pass(int D, const Leaf& leaf, u64 n_in, u64 n_out, u8* out, int level0) {
    for (u64 gid = 0; gid < n_out; gid++) {
        switch (D) {
            case 0: lane_pass<0, Leaf>(leaf, gid, n_in, out, zt(), level0); break;
            case 1: lane_pass<1, Leaf>(leaf, gid, n_in, out, zt(), level0); break;
            case 2: lane_pass<2, Leaf>(leaf, gid, n_in, out, zt(), level0); break;
            case 3: lane_pass<3, Leaf>(leaf, gid, n_in, out, zt(), level0); break;
            case 4: lane_pass<4, Leaf>(leaf, gid, n_in, out, zt(), level0); break;
            case 5: lane_pass<5, Leaf>(leaf, gid, n_in, out, zt(), level0); break;
            default: lane_pass<6, Leaf>(leaf, gid, n_in, out, zt(), level0); break;
        }
    }
}

template <class Leaf>
static Node tile_lane(const Leaf& leaf, u64 first, u64 n, int level0) {
    return Subtree<TILE_D, Leaf>::run(leaf, first, n, zt(), level0);
}

extern "C" {

void hs_zero_table(u8* out /* 65*32 */) {
    for (int d = 0; d <= 64; d++) node_store(zt()->z[d], out + 32 * d);
}

void hs_hash64(const u8* l, const u8* r, u8* out) {
    Node a, b;
    node_load(a, l);
    node_load(b, r);
    node_store(hash64(a, b), out);
}

void hs_sha256(const u8* data, u64 len, u8* out) {
    Sha256Stream s;
    sha256_init(s);
    sha256_update(s, data, len);
    u32 dg[8];
    sha256_final(s, dg);
    for (int k = 0; k < 8; k++) {
        out[4 * k] = (u8)(dg[k] >> 24);
        out[4 * k + 1] = (u8)(dg[k] >> 16);
        out[4 * k + 2] = (u8)(dg[k] >> 8);
        out[4 * k + 3] = (u8)dg[k];
    }
}

// kind: 0 chunks, 1 nodes, 2 validators, 3 bytes48, 4 pair64, 5 eth1data (LeafKind of merkle_driver.h)
// `in` may be an unaligned slice; out receives n_out = ceil(n_in / 2^D) nodes.
u64 hs_pass(int kind, int D, const u8* in, u64 in_bytes, u64 n_in, u8* out, int level0) {
    u64 n_out = (n_in + (1ull << D) - 1) >> D;
    switch (kind) {
        case 0: run_pass(D, ChunkLeaves{in, in_bytes}, n_in, n_out, out, level0); break;
        case 1: run_pass(D, NodeLeaves{in}, n_in, n_out, out, level0); break;
        case 2: run_pass(D, ValidatorLeaves{in, in_bytes}, n_in, n_out, out, level0); break;
        case 3: run_pass(D, Bytes48Leaves{in, in_bytes}, n_in, n_out, out, level0); break;
        case 4: run_pass(D, Pair64Leaves{in, in_bytes}, n_in, n_out, out, level0); break;
        case 6: run_pass(D, U64x2Leaves{in, in_bytes}, n_in, n_out, out, level0); break;
        case 7: run_pass(D, U64x3Leaves{in, in_bytes}, n_in, n_out, out, level0); break;
        default: run_pass(D, Eth1DataLeaves{in, in_bytes}, n_in, n_out, out, level0); break;
    }
    return n_out;
}

// One wave of the registry's staged leaf pass (merkle.hip k_merkle_pass<2, ValidatorLeaves>) with the kernel's addressing: the
// wave's 256 records start at `in + first_byte` (any alignment), each of the four steps copies the 485 aligned 16-byte vectors
// that cover its 64 records into a 7 760-byte stage -- vector 484 only where the step is misaligned, as the kernel loads it --,
// lane i hashes record 64 k + i through StagedRecord, the 256 roots go through the stage word-major in two halves, lane j
// stores the node over records 4 j .. 4 j + 3.  out: 64 nodes.  `in` must be readable 15 bytes before first_byte.
void hs_staged_validator_wave(const u8* in, u64 first_byte, u8* out) {
    const u8* src = in + first_byte;
    const u32 adj = (u32)((u64)src & 15);
    const u8* vsrc = src - adj;
    std::vector<u32> stage(4 * VAL_STAGE_VECS, 0xdeadbeefu);
    Node r[4][64];
    for (int k = 0; k < 4; k++) {
        const u8* g = vsrc + (u64)VAL_STEP_BYTES * k;
        for (u32 lane = 0; lane < 64; lane++)
            for (int i = 0; i < 8; i++) {
                const u32 j = lane + 64 * i;
                if (j >= VAL_STAGE_VECS) continue;
                if (j < VAL_STAGE_VECS - 1 || adj) std::memcpy(&stage[4 * j], g + 16ull * j, 16);
                else std::memset(&stage[4 * j], 0, 16);
            }
        for (u32 lane = 0; lane < 64; lane++) {
            StagedRecord rec = staged_record(stage.data(), adj, lane);
            r[k][lane] = validator_root_from_words(rec);
        }
    }
    Node c[4][64];
    for (int h = 0; h < 2; h++) {
        for (int k = 0; k < 4; k++)
            for (u32 lane = 0; lane < 64; lane++)
                for (int w = 0; w < 4; w++) stage[staged_root_dword(w, 64 * k + lane)] = r[k][lane].w[4 * h + w];
        for (u32 lane = 0; lane < 64; lane++)
            for (int w = 0; w < 4; w++)
                for (int q = 0; q < 4; q++) c[q][lane].w[4 * h + w] = stage[staged_root_dword(w, 4 * lane) + q];
    }
    for (u32 lane = 0; lane < 64; lane++) node_store(hash64(hash64(c[0][lane], c[1][lane]), hash64(c[2][lane], c[3][lane])), out + 32 * lane);
}

// the tile stage exactly as k_tree_tiles runs it (merkle.h TileDesc): per tile 256 lanes x Subtree<2>, then up to
// 8 levels pairwise with the virtual-pair shortcut; one node per tile at level min(top, level0 + 10)
u64 hs_tiles(int kind, const u8* in, u64 in_bytes, u64 n0, u32 level0, u32 top, u8* out) {
    const u64 n_tiles = (n0 + TILE_NODES - 1) / TILE_NODES;
    for (u64 tile = 0; tile < n_tiles; tile++) {
        std::vector<Node> nodes(TILE_LANES);
        for (u32 t = 0; t < TILE_LANES; t++) {
            const u64 first = tile * TILE_NODES + ((u64)t << TILE_D);
            switch (kind) {
                case 1: nodes[t] = tile_lane(NodeLeaves{in}, first, n0, (int)level0); break;
                case 3: nodes[t] = tile_lane(Bytes48Leaves{in, in_bytes}, first, n0, (int)level0); break;
                case 4: nodes[t] = tile_lane(Pair64Leaves{in, in_bytes}, first, n0, (int)level0); break;
                case 5: nodes[t] = tile_lane(Eth1DataLeaves{in, in_bytes}, first, n0, (int)level0); break;
                case 6: nodes[t] = tile_lane(U64x2Leaves{in, in_bytes}, first, n0, (int)level0); break;
                case 7: nodes[t] = tile_lane(U64x3Leaves{in, in_bytes}, first, n0, (int)level0); break;
                default: nodes[t] = tile_lane(ChunkLeaves{in, in_bytes}, first, n0, (int)level0); break;
            }
        }
        u32 lvl = level0 + TILE_D, m = TILE_LANES;
        while (lvl < top && m > 1) {
            const u32 pairs = m >> 1;
            std::vector<Node> h(pairs);
            for (u32 t = 0; t < pairs; t++) {
                const u64 left_first = tile * TILE_NODES + ((u64)(2 * t) << (lvl - level0));
                h[t] = left_first >= n0 ? zt()->z[lvl + 1] : hash64(nodes[2 * t], nodes[2 * t + 1]);
            }
            for (u32 t = 0; t < pairs; t++) nodes[t] = h[t];
            m = pairs;
            lvl++;
        }
        node_store(nodes[0], out + 32ull * tile);
    }
    return n_tiles;
}

// the finishing job exactly as k_tree_jobs runs it (level by level, zero-ladder climb, mix-in)
void hs_tree_job(const u8* in, u32 n, u32 level, u32 depth, int mix, u64 mix_len, u8* out) {
    std::vector<Node> nodes(n ? n : 1);
    for (u32 i = 0; i < n; i++) node_load(nodes[i], in + 32ull * i);
    u32 m = n, lvl = level;
    while (m > 1) {
        u32 pairs = (m + 1) >> 1;
        std::vector<Node> h(pairs);
        for (u32 i = 0; i < pairs; i++) {
            Node l = nodes[2 * i];
            Node r = (2 * i + 1 < m) ? nodes[2 * i + 1] : zt()->z[lvl];
            h[i] = hash64(l, r);
        }
        for (u32 i = 0; i < pairs; i++) nodes[i] = h[i];
        m = pairs;
        lvl++;
    }
    Node x = (n == 0) ? zt()->z[depth] : nodes[0];
    if (n != 0)
        for (; lvl < depth; lvl++) x = hash64(x, zt()->z[lvl]);
    if (mix) x = hash64(x, len_chunk(mix_len));
    node_store(x, out);
}


// merkleize exactly as merkle.hip::merkleize_device schedules it (schedule_merkleize is shared)
static void sim_merkleize(LeafKind kind, const u8* in, u64 in_bytes, u64 n0, u32 depth, bool mix, u64 mix_len,
                          u8* out, u64* hashes) {
    const MerkleSchedule sc = schedule_merkleize(kind, n0, depth, mix);
    std::vector<u8> a, b;
    const u8* cur = in;
    for (const PassStep& p : sc.passes) {
        std::vector<u8>& dst = (cur == a.data()) ? b : a;
        dst.assign(32 * (p.n_out ? p.n_out : 1), 0);
        if (p.first) hs_pass((int)kind, p.D, in, in_bytes, p.n_in, dst.data(), 0);
        else hs_pass(1, p.D, cur, 32 * p.n_in, p.n_in, dst.data(), (int)p.level_in);
        cur = dst.data();
    }
    if (sc.tile) {
        std::vector<u8>& dst = (cur == a.data()) ? b : a;
        dst.assign(32 * ((sc.tile_n_in + TILE_NODES - 1) / TILE_NODES + 1), 0);
        if (sc.tile_first) hs_tiles((int)kind, in, in_bytes, sc.tile_n_in, sc.tile_level_in, depth, dst.data());
        else hs_tiles(1, cur, 32 * sc.tile_n_in, sc.tile_n_in, sc.tile_level_in, depth, dst.data());
        cur = dst.data();
    }
    hs_tree_job(cur, sc.job_n, sc.job_level, depth, mix ? 1 : 0, mix_len, out);
    if (hashes) *hashes += sc.hashes;
}

// A resident field tree (csrc/state_tree.h) on the host: built over `before`, the encoding then becomes `after` (same entry count
// or a longer one with the same tree height), the touched entries are MARKED in the order given (duplicates allowed), the dirty
// list is CLIMBED in an order shuffled by `seed`, and the finishing job reduces level T to the root.  Threads run one after the
// other here: what is checked is the ticket logic -- exactly the dirty paths re-hashed, every counter back at zero -- not the
// memory model.  Returns the hash64 of the climbs; *left = counters / flags that are not zero afterwards (must be 0).
u64 hs_tree_update(int kind, const u8* before, u64 bytes_before, u64 n0_before, const u8* after, u64 bytes_after, u64 n0_after, u32 depth,
                   int mix, u64 mix_len, const u64* marks, u32 n_marks, u64 seed, u8* out_root, u32* left, u64* rebuild_hashes) {
    TreeGeom g{};
    g.kind = (u32)kind;
    g.H = ceil_log2_u64(n0_after ? n0_after : 1);
    g.T = tree_top_level(g.kind, g.H);
    const u64 cap = 1ull << g.H;
    std::vector<u8> lvl0(kind != LEAF_CHUNKS ? 32 * cap : 0), nodes(32 * cap);
    std::vector<u32> flag((cap + 31) / 32, 0), rcount(cap >> g.T, 0);
    std::vector<uint16_t> rlist(cap, 0);
    g.lvl0 = kind != LEAF_CHUNKS ? lvl0.data() : nullptr;
    g.nodes = nodes.data();
    g.flag0 = flag.data();
    g.rcount = rcount.data();
    g.rlist = rlist.data();
    g.src = before;
    g.bytes = bytes_before;
    g.n0 = n0_before;
    if (g.lvl0)
        for (u64 e = 0; e < g.n0; e++) node_store(tree_leaf(g, e), g.lvl0 + 32 * e);
    for (u32 k = 0; k < g.T;) {  // the rebuild launches of ResidentTrees::update
        const u32 D = g.T - k >= 3 ? 3 : g.T - k;
        const u64 n_out = tree_level_count(g.n0, k + D);
        for (u64 i = 0; i < n_out; i++) {
            if (D == 3) (void)TreeSpan<3>::run(g, k, i, zt());
            else if (D == 2) (void)TreeSpan<2>::run(g, k, i, zt());
            else (void)TreeSpan<1>::run(g, k, i, zt());
        }
        k += D;
    }
    if (rebuild_hashes) *rebuild_hashes = tree_rebuild_hashes(g);
    g.src = after;
    g.bytes = bytes_after;
    g.n0 = n0_after;
    std::vector<u32> active(TREE_ACTIVE_CAP);
    u32 n_active = 0;
    for (u32 i = 0; i < n_marks; i++) tree_mark(g, 3, marks[i], active.data(), &n_active);
    u64 x = seed | 1, hashes = 0;
    u32 bad = 0;
    std::vector<u32> lcnt(1u << g.T);
    for (u32 a = 0; a < n_active; a++) {  // one "workgroup" per active region (k_tree_climb)
        if ((active[a] >> 16) != 3) return ~0ull;
        const u32 region = active[a] & 0xffffu, n = g.rcount[region];
        std::vector<uint16_t> list(g.rlist + ((u64)region << g.T), g.rlist + ((u64)region << g.T) + n);
        std::fill(lcnt.begin(), lcnt.end(), 0u);
        for (u32 j = 0; j < n; j++) tree_region_count(g, lcnt.data(), list[j]);
        g.rcount[region] = 0;
        for (u32 i = n; i > 1; i--) {  // pass 2 in an order shuffled by a 64-bit LCG
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(list[i - 1], list[(x >> 33) % i]);
        }
        // k_tree_climb: every lane reads its path (counters as pass 1 left them, siblings as they are NOW) before any lane climbs --
        // a climb that trusted a stale sibling, or a counter another lane has taken down, shows up here
        std::vector<TreePath> paths(n);
        for (u32 j = 0; j < n; j++) tree_region_path(g, lcnt.data(), region, list[j], zt(), paths[j]);
        if (n > 64)
            for (u32 j = 0; j < n; j++) paths[j].both = 0xffffffffu;  // (the kernel's crowded-region rule: a ticket at every level)
        for (u32 j = 0; j < n; j++) hashes += tree_region_climb(g, lcnt.data(), region, list[j], zt(), paths[j]);
        for (u32 c : lcnt) bad += c != 0;
    }
    for (u32 f : flag) bad += f != 0;
    for (u32 c : rcount) bad += c != 0;
    if (left) *left = bad;
    const u8* in = g.T == 0 ? g.lvl0 : g.nodes + 32 * tree_heap_off(g.H, g.T);
    hs_tree_job(in, (u32)tree_level_count(g.n0, g.T), g.T, depth, mix, mix_len, out_root);
    return hashes;
}

u64 hs_merkleize(int kind, const u8* in, u64 in_bytes, u64 n0, u32 depth, int mix, u64 mix_len, u8* out) {
    u64 h = 0;
    sim_merkleize((LeafKind)kind, in, in_bytes, n0, depth, mix != 0, mix_len, out, &h);
    return h;
}

// state_deneb.hip::state_root_device on the lane simulator: same plan, same order (any fork from altair on; the payload
// header's extra_data offset is checked on the host copy like the host-pointer entry does)
int hs_state_root_fork(int fork, const u8* ssz, u64 n_bytes, int preset, u8* out, u64* hashes);
int hs_state_root_deneb(const u8* ssz, u64 n_bytes, int preset, u8* out, u64* hashes) {
    return hs_state_root_fork(FORK_DENEB, ssz, n_bytes, preset, out, hashes);
}
int hs_state_root_fork(int fork, const u8* ssz, u64 n_bytes, int preset, u8* out, u64* hashes) {
    StatePlan plan;
    if (preset < 0 || preset > 1 || fork < FORK_ALTAIR || fork > FORK_LAST) return -3;
    const FixedLayout L = layout_for(STATE_PRESETS[preset], fork);
    const u8* h_payload = nullptr;
    if (fork >= FORK_BELLATRIX && n_bytes >= L.size) {
        const u64 h = rd32(ssz + L.payload_header_off);
        if (h > n_bytes || n_bytes - h < payload_header_fixed(fork)) return -3;
        h_payload = ssz + h;
    }
    if (!build_state_plan(fork, ssz, n_bytes, preset, plan, nullptr, h_payload)) return -3;
    std::vector<u8> small(32ull * plan.n_small_chunks, 0);
    for (const GatherDesc& g : plan.gathers) {
        u32 d[8];
        u64 lim = g.src_off + g.n_bytes;
        if (lim > n_bytes) lim = n_bytes;
        load_bytes_le<8>(d, ssz, g.src_off, lim);
        if (g.last_and && g.n_bytes) {
            const u32 b = g.n_bytes - 1;
            d[b >> 2] &= ~(0xffu << (8 * (b & 3))) | ((g.last_and & 0xffu) << (8 * (b & 3)));
        }
        std::memcpy(small.data() + 32ull * g.dst_chunk, d, 32);
    }
    u64 hc = plan.small_hashes;
    for (const BigField& b : plan.bigs)
        sim_merkleize(b.kind, ssz + b.src, b.bytes, b.n0, b.depth, b.mix, b.mix_len, small.data() + 32ull * b.out_chunk, &hc);
    for (int l = 0; l < 3; l++)
        for (const TreeJob& j : plan.jobs[l])
            hs_tree_job(small.data() + j.in_off, j.n, j.level, j.depth, (int)j.mix, j.mix_len, small.data() + j.out_off);
    std::memcpy(out, small.data() + 32ull * plan.root_chunk, 32);
    if (hashes) *hashes = hc;
    return 0;
}

// shuffle.hip on the lane simulator: pivots, source table, one walk per index
void hs_shuffle(const u64* in, u64 n, const u8* seed32, u32 rounds, u64* out) {
    if (n == 0) return;
    ShuffleSeed sd;
    for (int i = 0; i < 8; i++)
        sd.w[i] = ((u32)seed32[4 * i] << 24) | ((u32)seed32[4 * i + 1] << 16) | ((u32)seed32[4 * i + 2] << 8) | seed32[4 * i + 3];
    const u64 nb = (n + 255) / 256;
    std::vector<u64> piv(rounds ? rounds : 1);
    std::vector<u32> table((size_t)rounds * nb * 8 + 8);
    for (u32 r = 0; r < rounds; r++) piv[r] = shuffle_pivot(sd, r, n);
    for (u64 i = 0; i < (u64)rounds * nb; i++) shuffle_hash_block(table.data() + i * 8, sd, (u32)(i / nb), true, (u32)(i % nb));
    for (u64 i = 0; i < n; i++) {
        const u64 j = shuffled_index(i, n, rounds, piv.data(), table.data(), nb);
        out[i] = in ? in[j] : j;
    }
}

// ssz_generic.hip::ecgpu_htr_ssz on the lane simulator: same plan, same order
int hs_htr_ssz(const ecgpu_ssz_type* types, u32 n_types, const u32* fields, u32 n_field_refs, u32 root_type, const u8* ssz, u64 n_bytes,
               u8* out, u64* hashes) {
    SszPlan plan;
    static const u8 empty[4] = {0, 0, 0, 0};
    if (!build_ssz_plan(types, n_types, fields, n_field_refs, root_type, ssz ? ssz : empty, n_bytes, plan)) return -3;
    std::vector<u8> small(32ull * plan.n_chunks, 0);
    for (const GatherDesc& g : plan.gathers) {
        u32 d[8];
        u64 lim = g.src_off + g.n_bytes;
        if (lim > n_bytes) lim = n_bytes;
        load_bytes_le<8>(d, ssz, g.src_off, lim);
        if (g.last_and && g.n_bytes) {
            const u32 b = g.n_bytes - 1;
            d[b >> 2] &= ~(0xffu << (8 * (b & 3))) | ((g.last_and & 0xffu) << (8 * (b & 3)));
        }
        std::memcpy(small.data() + 32ull * g.dst_chunk, d, 32);
    }
    u32 max_level = (u32)plan.jobs.size();
    for (auto& b : plan.bigs)
        if (b.level + 1 > max_level) max_level = b.level + 1;
    u64 hc = 0;
    for (u32 l = 1; l < max_level; l++) {
        for (auto& b : plan.bigs) {
            if (b.level != l) continue;
            std::vector<u8> src;
            const u8* in = ssz + b.src;
            if (b.kind == LEAF_NODES) {
                src.assign(small.begin() + 32ull * b.src, small.begin() + 32ull * (b.src + b.n0));
                in = src.data();
            }
            sim_merkleize(b.kind, in, b.bytes, b.n0, b.depth, b.mix, b.mix_len, small.data() + 32ull * b.out_chunk, &hc);
        }
        if (l < plan.jobs.size())
            for (const TreeJob& j : plan.jobs[l])
                hs_tree_job(small.data() + j.in_off, j.n, j.level, j.depth, (int)j.mix, j.mix_len, small.data() + j.out_off);
    }
    std::memcpy(out, small.data(), 32);
    if (hashes) *hashes = plan.hashes;
    return 0;
}

}  // extern "C"
