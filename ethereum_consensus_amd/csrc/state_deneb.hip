// hash_tree_root(BeaconState) for every fork the reference defines up to deneb (phase0 .. deneb; the file keeps its first
// name), driven from the state's SSZ encoding
// (plan: state_plan.h).  The host only walks SSZ offsets and emits descriptors; every hash64
// runs on the GPU.  Big fields go through the pass kernels straight from the device-resident
// encoding (read once); the ~60 small chunks are gathered into one staging buffer and reduced
// by three batched k_tree_jobs launches (leaf containers -> nested containers -> the 28-field
// state container).
#include <algorithm>
#include <cstring>
#include <vector>

#include "merkle_driver.h"
#include "state_fields.h"
#include "state_plan.h"
#include "state_tree_host.h"

#include <cstdlib>

namespace ecg {

// ---- one BeaconState over several GPUs (BASELINE north_star: "2^20-validator batch at 1, 2, 4 and 8 GPUs") ----
// The five registry-sized lists (validators, balances, the two participation lists, inactivity_scores: 99.2 % of a mainnet
// state's hash64) are cut into aligned power-of-two subtrees, one per rank; everything else (the 8 192-entry root vectors,
// randao mixes, sync committees, the small containers: 0.8 %) is computed redundantly by every rank.
//   phase A (ecgpu_beacon_state_shard_subroots_dev): rank g reduces ITS subtree of each of the five lists -> 5 nodes;
//   exchange: one all-gather of 5 x 32 bytes per rank (the caller's: RCCL in bench.py, host memory in a threaded host);
//   phase B (ecgpu_htr_beacon_state_sharded_dev): every rank finishes the five lists from the gathered nodes (they enter at
//   level log2(width): one finishing job each, zero ladder to the list limit, length mix-in) and computes the rest of the
//   state and the root -- one fused tail launch, the five jobs riding in it.
constexpr u32 N_SHARDED_LISTS = 5;
static int sharded_list_index(const BigField& b) {  // position among the five, or -1
    switch (b.out_chunk) {
        case 11: return b.kind == LEAF_VALIDATORS ? 0 : -1;
        case 12: return 1;
        case 15: return 2;
        case 16: return 3;
        case 21: return 4;
        default: return -1;
    }
}
// leaves per rank: the smallest power of two W with W * world >= n0 (ethereum_consensus_amd/shard.py subtree_width)
static u64 shard_width(u64 n0, u32 world) {
    const u64 per = n0 ? (n0 + world - 1) / world : 1;
    return 1ull << ceil_log2_u64(per);
}
struct ShardTop {
    const u8* d_all;   // world x 5 x 32 bytes, rank-major: what the all-gather of the phase-A outputs leaves on every rank
    u32 world;
    const u8* d_keep;  // may be null.  The field roots phase A left behind (64 x 32 bytes): phase B then hashes nothing but the
                       // five list tops and the state container -- the other fields ran underneath phase A's validator pass
};
constexpr u32 SHARD_FIELD_SLOTS[N_SHARDED_LISTS] = {11, 12, 15, 16, 21};

static int run_state_plan(hipStream_t s, ThreadCtx* c, const u8* d_ssz, u64 n_bytes, StatePlan& plan, int fork, bool dev_check,
                          u8* d_root, const ResidentTrees* trees, const u8* ext_roots, u8* d_field_roots, int* d_status, const u8* ext_src,
                          u64 ext_total, const u8* ext2_src = nullptr);

// trees != nullptr (resident state): every big field with a cached tree (state_tree.h: interior nodes kept on the device,
// brought up to date along dirty paths by the caller) enters the root as ONE finishing job over the <= 512 nodes of its
// tree's top cached level -- no pass, no tile stage: a mainnet root is ~7 k hash64 here instead of 10.1 M.
static int state_root_device(hipStream_t s, ThreadCtx* c, const u8* d_ssz, u64 n_bytes, const u8* h_fixed,
                             int preset, u8* d_root, const ResidentTrees* trees = nullptr, int fork = FORK_DENEB, const u8* ext_roots = nullptr,
                             const u8* h_payload_fixed = nullptr, u8* d_field_roots = nullptr, int* d_status = nullptr,
                             const ShardTop* shard = nullptr) {
    StatePlan plan;
    if (!build_state_plan(fork, h_fixed, n_bytes, preset, plan, ext_roots, h_payload_fixed)) {
        set_last_error(plan.error);
        return ECGPU_ERR_BAD_ARG;
    }
    if (shard) {
        // phase B: the five lists leave the plan; their sub-roots are gathered from the caller's buffer by the finishing job
        // that climbs them to the list limit
        std::vector<BigField> keep;
        for (const BigField& b : plan.bigs) {
            const int f = sharded_list_index(b);
            if (f < 0) {
                keep.push_back(b);
                continue;
            }
            const u64 W = shard_width(b.n0, shard->world);
            const u32 n_sub = (u32)((b.n0 + W - 1) / W);
            const u32 c0 = plan.n_small_chunks;
            plan.n_small_chunks += n_sub ? n_sub : 1;
            for (u32 r = 0; r < n_sub; r++) plan.gathers.push_back({32ull * (N_SHARDED_LISTS * r + (u32)f), 32u, c0 + r, 0u, 1u});
            TreeJob j;
            j.in_off = 32ull * c0;
            j.out_off = 32ull * b.out_chunk;
            j.n = n_sub;
            j.level = ceil_log2_u64(W);
            j.depth = b.depth;
            j.mix = b.mix ? 1 : 0;
            j.mix_len = b.mix_len;
            plan.jobs[0].push_back(j);
            u64 cnt = n_sub, l = j.level;
            while (cnt > 1) {
                cnt = (cnt + 1) / 2;
                plan.small_hashes += cnt;
                l++;
            }
            if (n_sub) plan.small_hashes += b.depth - l;
            if (b.mix) plan.small_hashes++;
        }
        plan.bigs.swap(keep);
        if (shard->d_keep) {
            // ... and with phase A's field roots handed in, every OTHER field leaves it too: their roots are copied from that
            // block into the field slots (one gathered chunk each), and what is left to hash is the five list tops above and
            // the state container
            StatePlan top;
            top.n_small_chunks = plan.n_small_chunks;
            top.root_chunk = plan.root_chunk;
            top.payload_header_off = plan.payload_header_off;
            for (const GatherDesc& g : plan.gathers)
                if (g.src_sel == 1) top.gathers.push_back(g);
            const u32 nf = state_field_count(fork);
            for (u32 f = 0; f < nf; f++) {
                bool deep = false;
                for (u32 k = 0; k < N_SHARDED_LISTS; k++) deep = deep || SHARD_FIELD_SLOTS[k] == f;
                if (!deep) top.gathers.push_back({32ull * f, 32u, f, 0u, 2u});
            }
            const size_t n_tops = N_SHARDED_LISTS;
            top.jobs[0].assign(plan.jobs[0].end() - n_tops, plan.jobs[0].end());
            top.jobs[2] = plan.jobs[2];
            top.small_hashes = 0;
            for (const TreeJob& j : top.jobs[0]) top.small_hashes += (j.n ? j.n - 1 + (j.depth - j.level) : 0) + j.mix;  // an upper bound
            top.small_hashes += state_field_chunks(fork) - 1;
            plan = top;
        }
    }
    const bool dev_check = fork >= FORK_BELLATRIX && !h_payload_fixed && plan.payload_header_off != ~0ull;
    return run_state_plan(s, c, d_ssz, n_bytes, plan, fork, dev_check, d_root, trees, ext_roots, d_field_roots, d_status,
                          shard ? shard->d_all : nullptr, shard ? 32ull * N_SHARDED_LISTS * shard->world : 0, shard ? shard->d_keep : nullptr);
}

// phase A of a sharded state root: this rank's subtree of each of the five lists -> d_subroots[5 x 32].  With d_keep (64 x 32
// bytes, the caller's) the rank also computes every OTHER field of the state in the same launches -- their tile stages and
// chains run underneath its validator pass -- and leaves the field roots there for phase B, which then only climbs the five
// list tops and hashes the container: what stands between the all-gather and the root is ~30 dependent hash64, not a state's tail.
static int state_shard_subroots_device(hipStream_t s, ThreadCtx* c, const u8* d_ssz, u64 n_bytes, const u8* h_fixed, int preset,
                                       int fork, u32 rank, u32 world, u8* d_subroots, u8* d_keep) {
    StatePlan full;
    if (!build_state_plan(fork, h_fixed, n_bytes, preset, full, nullptr, nullptr)) {
        set_last_error(full.error);
        return ECGPU_ERR_BAD_ARG;
    }
    auto restricted = [&](const BigField& b, u32 out_chunk) {
        const u64 W = shard_width(b.n0, world), rec = leaf_record_bytes(b.kind);
        const u64 lo = std::min<u64>((u64)rank * W, b.n0), hi = std::min<u64>(((u64)rank + 1) * W, b.n0);
        const u64 lo_b = std::min<u64>(lo * rec, b.bytes), hi_b = std::min<u64>(hi * rec, b.bytes);  // a packed list's last chunk may be partial
        return BigField{b.kind, b.src + lo_b, hi_b - lo_b, hi - lo, ceil_log2_u64(W), false, 0, out_chunk};
    };
    u8*& sc_slot = c->small_scratch[s];  // one per stream: two phase-A calls of a thread on two streams must not share it
    if (!sc_slot) ECG_HIP_CHECK(hipMalloc((void**)&sc_slot, 4096));
    u8* d_sc = sc_slot;  // <= 2 KB field-root block + the 32-byte root of a container nobody reads
    if (d_keep) {
        // the whole state plan, the five lists cut down to this rank's subtrees (their field slots then hold SUB-roots, the
        // container job a root nobody reads)
        for (BigField& b : full.bigs)
            if (sharded_list_index(b) >= 0) b = restricted(b, b.out_chunk);
        full.small_hashes = 0;  // (work accounting of this form: the big fields only)
        int rc = run_state_plan(s, c, d_ssz, n_bytes, full, fork, false, d_sc + 2048, nullptr, nullptr, d_keep, nullptr, nullptr, 0);
        if (rc) return rc;
        for (u32 k = 0; k < N_SHARDED_LISTS; k++)
            ECG_HIP_CHECK(hipMemcpyAsync(d_subroots + 32 * k, d_keep + 32 * SHARD_FIELD_SLOTS[k], 32, hipMemcpyDeviceToDevice, s));
        return ECGPU_SUCCESS;
    }
    StatePlan plan;
    for (const BigField& b : full.bigs) {
        const int f = sharded_list_index(b);
        if (f >= 0) plan.bigs.push_back(restricted(b, (u32)f));
    }
    // the fused tail ends in a container job: here a 5-leaf tree over the sub-roots, whose root nobody reads
    Builder B;
    const u32 root_chunk = B.alloc(1);
    B.job(2, 0, N_SHARDED_LISTS, 3, root_chunk);
    plan.jobs[2] = B.jobs[2];
    plan.n_small_chunks = B.next_chunk;
    plan.root_chunk = root_chunk;
    plan.small_hashes = B.hashes;
    // the field-root chunks of the small buffer are handed back whole (d_field_roots); the first five are the sub-roots
    int rc = run_state_plan(s, c, d_ssz, n_bytes, plan, fork, false, d_sc + 2048, nullptr, nullptr, d_sc, nullptr, nullptr, 0);
    if (rc) return rc;
    ECG_HIP_CHECK(hipMemcpyAsync(d_subroots, d_sc, 32 * N_SHARDED_LISTS, hipMemcpyDeviceToDevice, s));
    return ECGPU_SUCCESS;
}

static int run_state_plan(hipStream_t s, ThreadCtx* c, const u8* d_ssz, u64 n_bytes, StatePlan& plan, int fork, bool dev_check,
                          u8* d_root, const ResidentTrees* trees, const u8* ext_roots, u8* d_field_roots, int* d_status, const u8* ext_src,
                          u64 ext_total, const u8* ext2_src) {
    // resident state: the cached fields leave the plan; each comes back further down as ONE unit of the tail
    std::vector<std::pair<u32, u32>> cached;  // (tree slot = position in the plan, field-root chunk)
    if (trees) {
        std::vector<BigField> keep;
        for (size_t i = 0; i < plan.bigs.size(); i++) {
            if (i < TREE_MAX_FIELDS && trees->f[i].live) cached.push_back({(u32)i, plan.bigs[i].out_chunk});
            else keep.push_back(plan.bigs[i]);
        }
        plan.bigs.swap(keep);
    }
    std::vector<const u8*> fptr(plan.bigs.size());
    for (size_t i = 0; i < plan.bigs.size(); i++) fptr[i] = d_ssz + plan.bigs[i].src;
    // ---- device buffers --------------------------------------------------------------------------
    // Everything the root needs from the host -- the tail's plan, its zeroed tickets, the gather descriptors and the (zeroed)
    // small-chunk buffer -- is ONE block, uploaded by one copy from a pinned slot in front of the passes.  (Round 3 first put
    // the five small operations of the previous schedule -- two descriptor uploads, two memsets, the gather launch -- on an
    // auxiliary stream under the passes: 1.09 -> 1.07 ms when nothing else had created streams, 1.19 ms in a process that had,
    // where the auxiliary stream shares a hardware queue with the caller's: the driver's bench.  One stream, one copy, no
    // gather launch -- the tail's units fetch the chunks they hash -- does not depend on that.)
    Arena& ar = c->arena(s);
    ar.reset();
    size_t need = 8192;
    for (auto& b : plan.bigs) need += merkle_ws_bytes(b.n0) + 512;
    const size_t small_bytes = 32ull * plan.n_small_chunks;
    const size_t n_counters = TAIL_MAX_FIELDS + TAIL_MAX_JOBS1 + 4;
    auto up256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t off_counters = up256(sizeof(TailPlan)), off_gath = off_counters + up256(n_counters * sizeof(u32)),
                 off_small = off_gath + up256(plan.gathers.size() * sizeof(GatherDesc)), block_bytes = off_small + up256(small_bytes);
    need += block_bytes + 2048;
    int rc = ar.reserve(need);
    if (rc) return rc;
    u8* d_block = ar.take(block_bytes);
    if (!d_block) return ECGPU_ERR_OOM;
    TailPlan* d_tail = (TailPlan*)d_block;
    u32* d_counters = (u32*)(d_block + off_counters);  // tickets: fields, nested containers, the state container
    GatherDesc* d_gath = (GatherDesc*)(d_block + off_gath);
    u8* d_small = d_block + off_small;
    // The device entry never sees the payload header on the host: the extra_data offset word the host entries check
    // (state_plan.h) is compared on the device (dev_check), and a mismatch poisons the root (32 x 0xFF) and sets *d_status.
    u64 hc = plan.small_hashes;
    for (const auto& sc : cached) {  // finishing jobs over the cached trees' top levels (offsets relative to the small-chunk buffer)
        plan.jobs[0].push_back(trees->job(sc.first, d_small, 32ull * sc.second));
        hc += trees->job_hashes(sc.first);
    }
    // Schedule (round 3): ONE stream.  The plan first (every tree is DESCRIBED here and launched further down), then the wide
    // passes -- the validator registry, 93 % of the hashes, and whatever other field is too wide for a tile stage --, then ONE launch for everything that is left: the tile stages of all fields, their
    // finishing jobs, the leaf containers, the nested containers and the state container, chained by arrival tickets inside the
    // kernel (merkle_driver.h TailPlan).  Round 2 overlapped the other 13 fields with the validator pass on two auxiliary streams:
    // their latency-bound workgroups cost the chip-filling pass 25 % (0.62 -> 0.78 ms) and the tail was ~6 dependent launches
    // joined by events.
    size_t biggest = 0;
    for (size_t i = 1; i < plan.bigs.size(); i++)
        if (plan.bigs[i].bytes > plan.bigs[biggest].bytes) biggest = i;
    static TailPlan tp_init{};
    TailPlan tp = tp_init;
    std::vector<size_t> order;  // (a plan may have no big field at all: phase B of a sharded root with the field roots handed in)
    if (!plan.bigs.empty()) order.push_back(biggest);
    for (size_t i = 0; i < plan.bigs.size(); i++)
        if (i != biggest) order.push_back(i);
    std::vector<u8*> wss;
    for (size_t i : order) {
        const BigField& b = plan.bigs[i];
        u8* ws = ar.take(merkle_ws_bytes(b.n0));
        if (!ws) return ECGPU_ERR_OOM;
        TreeJob dj;
        TileDesc td;
        u32 n_tiles = 0;
        wss.push_back(ws);
        rc = merkleize_device(s, b.kind, fptr[i], b.bytes, b.n0, b.depth, b.mix, b.mix_len, d_small + 32ull * b.out_chunk, ws, &hc, &dj, ar.base,
                              i != biggest, nullptr, &td, &n_tiles, MERKLEIZE_DESCRIBE);
        if (rc) return rc;
        if (n_tiles) {
            if (tp.n_fields >= TAIL_MAX_FIELDS) return ECGPU_ERR_BAD_ARG;
            td.first_wg = tp.n_tile_wgs;
            tp.fields[tp.n_fields++] = TailField{td, dj, n_tiles, TAIL_NONE, i == biggest ? 1u : 0u, 0u};
            tp.n_tile_wgs += n_tiles;
        } else {
            if (tp.n_jobs0 >= TAIL_MAX_JOBS0) return ECGPU_ERR_BAD_ARG;
            tp.jobs0[tp.n_jobs0] = dj;
            tp.jobs0_feeds[tp.n_jobs0++] = TAIL_NONE;
        }
    }
    // level jobs were planned relative to the small-chunk buffer: rebase them onto the arena like the deferred ones
    const u64 small_off = (u64)(d_small - ar.base);
    auto rebased = [&](TreeJob j) {
        j.in_off += small_off;
        j.out_off += small_off;
        return j;
    };
    if (tp.n_jobs0 + plan.jobs[0].size() + plan.jobs[1].size() > TAIL_MAX_JOBS0 || plan.jobs[1].size() > TAIL_MAX_JOBS1 || plan.jobs[2].size() != 1) {
        set_last_error("state plan does not fit the fused tail");
        return ECGPU_ERR_BAD_ARG;
    }
    for (const TreeJob& j : plan.jobs[0]) {
        tp.jobs0[tp.n_jobs0] = rebased(j);
        tp.jobs0_feeds[tp.n_jobs0++] = TAIL_NONE;
    }
    // which nested container (if any) a unit's root is an input of: its output chunk lies inside that container's input block
    std::vector<TreeJob> nested;
    for (const TreeJob& j : plan.jobs[1]) nested.push_back(rebased(j));
    std::vector<u32> deps(nested.size(), 0);
    auto feeds_of = [&](const TreeJob& unit) -> u32 {
        for (size_t k = 0; k < nested.size(); k++)
            if (unit.out_off >= nested[k].in_off && unit.out_off < nested[k].in_off + 32ull * nested[k].n) return (u32)k;
        return TAIL_NONE;
    };
    for (u32 f = 0; f < tp.n_fields; f++) {
        tp.fields[f].feeds = feeds_of(tp.fields[f].job);
        if (tp.fields[f].feeds != TAIL_NONE) deps[tp.fields[f].feeds]++;
    }
    for (u32 j = 0; j < tp.n_jobs0; j++) {
        tp.jobs0_feeds[j] = feeds_of(tp.jobs0[j]);
        if (tp.jobs0_feeds[j] != TAIL_NONE) deps[tp.jobs0_feeds[j]]++;
    }
    // a nested container nobody feeds (all of its inputs are gathered chunks) is a unit of its own; the others keep their order
    std::vector<u32> remap(nested.size(), TAIL_NONE);
    for (size_t k = 0; k < nested.size(); k++) {
        if (deps[k] == 0) {
            tp.jobs0[tp.n_jobs0] = nested[k];
            tp.jobs0_feeds[tp.n_jobs0++] = TAIL_NONE;
        } else {
            remap[k] = tp.n_jobs1;
            tp.jobs1[tp.n_jobs1] = nested[k];
            tp.jobs1_deps[tp.n_jobs1++] = deps[k];
        }
    }
    u32 direct = 0;
    for (u32 f = 0; f < tp.n_fields; f++) {
        if (tp.fields[f].feeds != TAIL_NONE) tp.fields[f].feeds = remap[tp.fields[f].feeds];
        if (tp.fields[f].feeds == TAIL_NONE) direct++;
    }
    for (u32 j = 0; j < tp.n_jobs0; j++) {
        if (tp.jobs0_feeds[j] != TAIL_NONE) tp.jobs0_feeds[j] = remap[tp.jobs0_feeds[j]];
        if (tp.jobs0_feeds[j] == TAIL_NONE) direct++;
    }
    tp.job2 = rebased(plan.jobs[2][0]);
    tp.final_parties = direct + tp.n_jobs1;
    tp.root_off = small_off + 32ull * plan.root_chunk;
    tp.froots_off = small_off;  // the first chunks of the small buffer are the roots of the state's fields (proofs: ssz_proof.hip)
    tp.n_froots = state_field_chunks(fork);
    tp.d_root = d_root;
    tp.d_field_roots = d_field_roots;
    tp.counters = d_counters;
    tp.src = d_ssz;
    tp.src_total = n_bytes;
    tp.gathers = d_gath;
    tp.n_gathers = (u32)plan.gathers.size();
    tp.small_off = small_off;
    tp.small_end = small_off + small_bytes;
    tp.chk_off = dev_check ? plan.payload_header_off + PAYLOAD_EXTRA_DATA_OFFSET_WORD : ~0ull;
    tp.chk_expect = (u32)payload_header_fixed(fork);
    tp.ext_src = ext_src;
    tp.ext_total = ext_total;
    tp.ext2_src = ext2_src;
    tp.d_status = d_status;
    // the block: plan, zero tickets, descriptors, zero chunks (+ phase0: the two list roots the generic planner computed)
    u8* h_block;
    hipEvent_t copied;
    rc = c->uploads.acquire(block_bytes, &h_block, &copied);
    if (rc) return rc;
    memset(h_block, 0, block_bytes);
    memcpy(h_block, &tp, sizeof(TailPlan));
    if (!plan.gathers.empty()) memcpy(h_block + off_gath, plan.gathers.data(), plan.gathers.size() * sizeof(GatherDesc));
    for (const StatePlan::ExtChunk& e : plan.ext_chunks) memcpy(h_block + off_small + 32ull * e.dst_chunk, ext_roots + e.src_off, 32);
    ECG_HIP_CHECK(hipMemcpyAsync(d_block, h_block, block_bytes, hipMemcpyHostToDevice, s));
    ECG_HIP_CHECK(hipEventRecord(copied, s));
    // the wide passes, then the one launch for everything that is left
    for (size_t k = 0; k < order.size(); k++) {
        const BigField& b = plan.bigs[order[k]];
        TreeJob dj;
        TileDesc td;
        u32 n_tiles = 0;
        rc = merkleize_device(s, b.kind, fptr[order[k]], b.bytes, b.n0, b.depth, b.mix, b.mix_len, d_small + 32ull * b.out_chunk, wss[k], nullptr, &dj,
                              ar.base, order[k] != biggest, nullptr, &td, &n_tiles, MERKLEIZE_LAUNCH);
        if (rc) return rc;
    }
    rc = launch_state_tail(s, d_tail, tp.n_tile_wgs + tp.n_jobs0, ar.base);
    if (rc) return rc;
    c->last_hash64 = hc;
    return ECGPU_SUCCESS;
}

// patch scatter: one workgroup per patch, byte copies (patches are small: a balance, a flag byte, a root)
struct PatchDesc {
    u64 dst_off, src_off, len;
};
__global__ void k_apply_patches(u8* state, const u8* data, const PatchDesc* p) {
    const PatchDesc d = p[blockIdx.x];
    for (u64 i = threadIdx.x; i < d.len; i += blockDim.x) state[d.dst_off + i] = data[d.src_off + i];
}

}  // namespace ecg

struct ResidentSink;  // the device-resident encoding as state_fields.h's FieldWriter sees it (below)

struct ecgpu_resident_state {
    ecg::FieldWriter<ResidentSink> queue;  // field-addressed writes / pushes not yet applied (round 6)
    int preset = 0;
    int fork = ecg::FORK_DENEB;
    u8* d_ssz = nullptr;   // encoding (cap_bytes allocated: lists grow in place, ecgpu_resident_state_append)
    u64 n_bytes = 0, cap_bytes = 0;
    u8* d_rootbuf = nullptr;  // 64 bytes: where the host-pointer root entry leaves its result
    std::vector<u8> h_fixed;  // host mirror of the fixed-size part (offsets and small fields): what the plan reads
    // every interior node of every big field's tree stays on the device (csrc/state_tree.h); a patch marks
    // the level-0 entries it touches, a root re-hashes their paths and nothing else
    ecg::ResidentTrees trees;
    u64 last_climb_hashes = 0;  // hash64 the dirty-path climbs of the last host-pointer root performed (device counter)
    hipEvent_t patched = nullptr;  // recorded behind the last patch / length change on the stream it ran on
    hipStream_t patch_stream = nullptr;
    hipEvent_t rooted = nullptr;   // recorded behind the last root: a patch on another stream waits for it before it overwrites bytes
    hipStream_t root_stream = nullptr;
    // phase0: the roots of previous / current_epoch_attestations (lists of variable-size elements: rooted through the generic
    // planner when the state is created and whenever ecgpu_resident_state_replace hands a list over -- on the host side of
    // the call, where its bytes are; the state root takes the two nodes as they are)
    u8 att_roots[64] = {};
    const u8* ext_roots() const { return fork == ecg::FORK_PHASE0 ? att_roots : nullptr; }
};

using namespace ecg;

extern "C" {

static int pending_attestations_root(const u8* ssz, u64 n_bytes, int preset, u8 root[32]);

int ecgpu_resident_state_create(int preset, const uint8_t* ssz, uint64_t n_bytes, ecgpu_resident_state_t** out) {
    return ecgpu_resident_state_create_fork(ECGPU_FORK_DENEB, preset, ssz, n_bytes, out);
}

int ecgpu_resident_state_create_fork(int fork, int preset, const uint8_t* ssz, uint64_t n_bytes, ecgpu_resident_state_t** out) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!ssz || !out || preset < 0 || preset > 1 || fork < FORK_PHASE0 || fork > FORK_LAST ||
        n_bytes < layout_for(STATE_PRESETS[preset], fork).size)
        return ECGPU_ERR_BAD_ARG;
    u8 att[64] = {};
    if (fork == FORK_PHASE0) {  // (round 5) the two PendingAttestation lists: rooted here, from the caller's bytes
        const FixedLayout L = layout_for(STATE_PRESETS[preset], fork);
        const u64 a = rd32(ssz + L.prev_attestations_off), b = rd32(ssz + L.cur_attestations_off);
        if (a > b || b > n_bytes || a < L.size) {
            set_last_error("SSZ offsets not monotonic");
            return ECGPU_ERR_BAD_ARG;
        }
        if ((rc = pending_attestations_root(ssz + a, b - a, preset, att))) return rc;
        if ((rc = pending_attestations_root(ssz + b, n_bytes - b, preset, att + 32))) return rc;
    }
    StatePlan plan;
    if (!build_state_plan(fork, ssz, n_bytes, preset, plan, fork == FORK_PHASE0 ? att : nullptr,
                          fork >= FORK_BELLATRIX ? ssz + rd32(ssz + layout_for(STATE_PRESETS[preset], fork).payload_header_off) : nullptr)) {
        set_last_error(plan.error);
        return ECGPU_ERR_BAD_ARG;
    }
    ecgpu_resident_state* st = new ecgpu_resident_state();
    std::memcpy(st->att_roots, att, 64);
    st->preset = preset;
    st->fork = fork;
    st->n_bytes = n_bytes;
    st->h_fixed.assign(ssz, ssz + layout_for(STATE_PRESETS[preset], fork).size);
    st->cap_bytes = n_bytes + (n_bytes >> 6) + (1u << 16);  // room for ~1.5 % growth before the buffer is reallocated
    ECG_HIP_CHECK(hipMalloc((void**)&st->d_ssz, st->cap_bytes));
    ECG_HIP_CHECK(hipMalloc((void**)&st->d_rootbuf, 64));
    ECG_HIP_CHECK(hipMemcpy(st->d_ssz, ssz, n_bytes, hipMemcpyHostToDevice));
    rc = st->trees.sync_geometry(plan);  // allocates the trees; every field is built by the first root
    if (rc) {
        ecgpu_resident_state_destroy(st);
        return rc;
    }
    *out = st;
    return ECGPU_SUCCESS;
}

// ---- lists that change length (add_validator_to_registry, phase0/block_processing.rs:317-349; the eth1_data_votes reset of
// process_eth1_data_reset; historical_summaries growing once per period) ------------------------------------------------------
namespace {
// Cross-stream ordering (advisor, round 5).  A state may be handed from one host thread to another (each has its own stream)
// or rooted on a caller's stream: whatever changes bytes or tree marks on stream `s` first waits for the last change and for
// the last root enqueued on OTHER streams -- stream order covers the same stream.
static int order_after_earlier_work(ecgpu_resident_state* st, hipStream_t s) {
    if (st->patched && st->patch_stream != s) ECG_HIP_CHECK(hipStreamWaitEvent(s, st->patched, 0));
    if (st->rooted && st->root_stream != s) ECG_HIP_CHECK(hipStreamWaitEvent(s, st->rooted, 0));
    return ECGPU_SUCCESS;
}
static int record_change(ecgpu_resident_state* st, hipStream_t s) {
    if (!st->patched) ECG_HIP_CHECK(hipEventCreateWithFlags(&st->patched, hipEventDisableTiming));
    ECG_HIP_CHECK(hipEventRecord(st->patched, s));
    st->patch_stream = s;
    return ECGPU_SUCCESS;
}
// variable-size fields of the state in encoding order: position of the offset word in the fixed part, element size
struct VarField {
    u64 word;
    u32 elem;
};
constexpr int N_VAR_FIELDS = 12;  // (9 .. 11: electra's three lists of pending operations, electra/beacon_state.rs:133-137)
static int var_fields(const ecgpu_resident_state* st, VarField out[N_VAR_FIELDS]) {
    const FixedLayout L = layout_for(STATE_PRESETS[st->preset], st->fork);
    // (slots 4 and 5 of a phase0 state: the PendingAttestation lists, element size 0 = variable -- ecgpu_resident_state_replace only)
    const bool p0 = st->fork == FORK_PHASE0;
    const VarField f[N_VAR_FIELDS] = {{L.historical_roots_off, 32}, {L.eth1_data_votes_off, 72}, {L.validators_off, 121}, {L.balances_off, 8},
                                      {p0 ? L.prev_attestations_off : L.prev_participation_off, p0 ? 0u : 1u},
                                      {p0 ? L.cur_attestations_off : L.cur_participation_off, p0 ? 0u : 1u}, {L.inactivity_scores_off, 8},
                                      {L.payload_header_off, 0}, {L.historical_summaries_off, 64}, {L.pending_balance_deposits_off, 16},
                                      {L.pending_partial_withdrawals_off, 24}, {L.pending_consolidations_off, 16}};
    for (int i = 0; i < N_VAR_FIELDS; i++) out[i] = f[i];
    return N_VAR_FIELDS;
}
static void wr32(u8* p, u32 v) {
    p[0] = (u8)v, p[1] = (u8)(v >> 8), p[2] = (u8)(v >> 16), p[3] = (u8)(v >> 24);
}
// replace bytes [pos, pos + remove) of the encoding by `insert_len` bytes (host pointer); the tail moves on the device
static int splice(ecgpu_resident_state* st, hipStream_t s, Arena& ar, u64 pos, u64 remove, const u8* insert, u64 insert_len) {
    const u64 tail = st->n_bytes - (pos + remove), new_bytes = st->n_bytes - remove + insert_len;
    if (new_bytes > 0xffffffffull) return ECGPU_ERR_BAD_ARG;  // SSZ offsets are 32 bits
    if (new_bytes > st->cap_bytes) {
        const u64 cap = new_bytes + (new_bytes >> 5) + (1u << 16);
        u8* nb = nullptr;
        ECG_HIP_CHECK(hipMalloc((void**)&nb, cap));
        ECG_HIP_CHECK(hipMemcpyAsync(nb, st->d_ssz, pos, hipMemcpyDeviceToDevice, s));
        if (tail) ECG_HIP_CHECK(hipMemcpyAsync(nb + pos + insert_len, st->d_ssz + pos + remove, tail, hipMemcpyDeviceToDevice, s));
        ECG_HIP_CHECK(hipStreamSynchronize(s));
        ECG_HIP_CHECK(hipFree(st->d_ssz));
        st->d_ssz = nb;
        st->cap_bytes = cap;
    } else if (tail && remove != insert_len) {
        ar.reset();
        int rc = ar.reserve(tail + 4096);  // overlapping move: through a temporary
        if (rc) return rc;
        u8* tmp = ar.take(tail);
        ECG_HIP_CHECK(hipMemcpyAsync(tmp, st->d_ssz + pos + remove, tail, hipMemcpyDeviceToDevice, s));
        ECG_HIP_CHECK(hipMemcpyAsync(st->d_ssz + pos + insert_len, tmp, tail, hipMemcpyDeviceToDevice, s));
    }
    if (insert_len) ECG_HIP_CHECK(hipMemcpyAsync(st->d_ssz + pos, insert, insert_len, hipMemcpyHostToDevice, s));
    st->n_bytes = new_bytes;
    return ECGPU_SUCCESS;
}
// change the byte length of variable field `fi` to new_len (append `data` when it grows, drop the tail when it shrinks)
enum ResizeMode { RESIZE_APPEND, RESIZE_TRUNCATE, RESIZE_REPLACE };
static int resize_field(ecgpu_resident_state* st, int fi, const u8* data, u64 add_len, u64 new_len_or_keep, ResizeMode mode) {
    const bool truncate = mode == RESIZE_TRUNCATE, replace = mode == RESIZE_REPLACE;
    VarField vf[N_VAR_FIELDS];
    var_fields(st, vf);
    const bool attestations = st->fork == FORK_PHASE0 && (fi == 4 || fi == 5);
    const bool header = fi == 7;  // latest_execution_payload_header: one container whose extra_data changes length (replace only)
    if (fi < 0 || fi >= N_VAR_FIELDS || vf[fi].word == NO_FIELD || (vf[fi].elem == 0 && !(replace && (attestations || header)))) {
        set_last_error(attestations ? "a list of variable-size elements changes through ecgpu_resident_state_replace" : "not a variable-length list of this fork");
        return ECGPU_ERR_BAD_ARG;
    }
    if (header) {  // what the reference's deserializer accepts: the fixed part, <= 32 bytes of extra_data, its offset word == the fixed size
        const u64 hf = payload_header_fixed(st->fork);
        if (!data || add_len < hf || add_len > hf + 32 || rd32(data + PAYLOAD_EXTRA_DATA_OFFSET_WORD) != hf) {
            set_last_error("payload header: bad length or extra_data offset");
            return ECGPU_ERR_BAD_ARG;
        }
    }
    u8 new_att_root[32];
    if (attestations) {  // the list's root, from the caller's bytes (which also validates the encoding), before anything moves
        int rc_att = pending_attestations_root(data ? data : (const u8*)"", add_len, st->preset, new_att_root);
        if (rc_att) return rc_att;
    }
    const u64 start = rd32(st->h_fixed.data() + vf[fi].word);
    u64 end = st->n_bytes;
    for (int k = fi + 1; k < N_VAR_FIELDS; k++)
        if (vf[k].word != NO_FIELD) {
            end = rd32(st->h_fixed.data() + vf[k].word);
            break;
        }
    const u64 cur = end - start;
    u64 pos, remove, insert;
    if (truncate) {
        if (new_len_or_keep > cur || new_len_or_keep % vf[fi].elem) return ECGPU_ERR_BAD_ARG;
        pos = start + new_len_or_keep, remove = cur - new_len_or_keep, insert = 0;
    } else if (replace) {
        if ((vf[fi].elem && add_len % vf[fi].elem) || (!data && add_len)) return ECGPU_ERR_BAD_ARG;
        pos = start, remove = cur, insert = add_len;
    } else {
        if (add_len % vf[fi].elem || (!data && add_len)) return ECGPU_ERR_BAD_ARG;
        pos = end, remove = 0, insert = add_len;
    }
    if (remove == 0 && insert == 0) return ECGPU_SUCCESS;
    const int64_t delta = (int64_t)insert - (int64_t)remove;
    {  // the resized state must still be a valid state of this fork (a list past its limit is not): checked on a copy of the
       // fixed part BEFORE anything moves, so a refused call leaves the resident state as it was
        std::vector<u8> probe(st->h_fixed);
        for (int k = fi + 1; k < N_VAR_FIELDS; k++)
            if (vf[k].word != NO_FIELD) wr32(probe.data() + vf[k].word, (u32)((int64_t)rd32(probe.data() + vf[k].word) + delta));
        StatePlan would_be;
        u8 ext_probe[64];
        std::memcpy(ext_probe, st->att_roots, 64);
        if (attestations) std::memcpy(ext_probe + 32 * (fi - 4), new_att_root, 32);
        if (!build_state_plan(st->fork, probe.data(), (u64)((int64_t)st->n_bytes + delta), st->preset, would_be,
                              st->fork == FORK_PHASE0 ? ext_probe : nullptr, nullptr)) {
            set_last_error(would_be.error);
            return ECGPU_ERR_BAD_ARG;
        }
    }
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    Arena& ar = c->arena(s);
    int rc = order_after_earlier_work(st, s);
    if (rc) return rc;
    rc = splice(st, s, ar, pos, remove, data, insert);
    if (rc) return rc;
    // later fields start `insert - remove` bytes later: their offset words change on the host mirror and on the device
    for (int k = fi + 1; k < N_VAR_FIELDS; k++)
        if (vf[k].word != NO_FIELD) {
            const u32 v = (u32)((int64_t)rd32(st->h_fixed.data() + vf[k].word) + delta);
            wr32(st->h_fixed.data() + vf[k].word, v);
            ECG_HIP_CHECK(hipMemcpyAsync(st->d_ssz + vf[k].word, st->h_fixed.data() + vf[k].word, 4, hipMemcpyHostToDevice, s));
        }
    // the trees follow: every field's offset may have moved; the resized field's new entries are marked dirty (a field whose
    // height changed, or that shrank, is rebuilt at the next root)
    if (attestations) std::memcpy(st->att_roots + 32 * (fi - 4), new_att_root, 32);
    StatePlan plan;
    if (!build_state_plan(st->fork, st->h_fixed.data(), st->n_bytes, st->preset, plan, st->ext_roots(), nullptr)) {
        set_last_error(plan.error);
        return ECGPU_ERR_BAD_ARG;
    }
    static const u32 field_chunk[N_VAR_FIELDS] = {7, 9, 11, 12, 15, 16, 21, 0, 27, 34, 35, 36};  // var_fields order -> field-root chunk of the list
    u32 slot = TREE_MAX_FIELDS;
    for (u32 k = 0; k < plan.bigs.size() && k < TREE_MAX_FIELDS; k++)
        if (plan.bigs[k].out_chunk == field_chunk[fi] && plan.bigs[k].mix) slot = k;
    const u64 old_bytes = slot < TREE_MAX_FIELDS ? st->trees.f[slot].g.bytes : 0;
    const bool was_live = slot < TREE_MAX_FIELDS && st->trees.f[slot].live;
    const u32 old_H = was_live ? st->trees.f[slot].g.H : 0;
    rc = st->trees.sync_geometry(plan);
    if (rc) return rc;
    if (was_live && st->trees.f[slot].live && !st->trees.f[slot].all_dirty && st->trees.f[slot].g.H == old_H) {
        FieldTree& t = st->trees.f[slot];
        if (truncate || replace) {
            t.all_dirty = true;
        } else {
            const u64 rec = leaf_record_bytes((LeafKind)t.g.kind);
            std::vector<u64> pairs;
            st->trees.collect_entries(slot, old_bytes / rec, (t.g.bytes - 1) / rec, pairs);
            if (!pairs.empty()) {
                ar.reset();
                rc = ar.reserve(8 * pairs.size() + 4096);
                if (rc) return rc;
                u64* d_pairs = (u64*)ar.take(8 * pairs.size());
                ECG_HIP_CHECK(hipMemcpyAsync(d_pairs, pairs.data(), 8 * pairs.size(), hipMemcpyHostToDevice, s));
                rc = st->trees.mark(s, d_pairs, (u32)pairs.size());
                if (rc) return rc;
            }
        }
    }
    if ((rc = record_change(st, s))) return rc;
    ECG_HIP_CHECK(hipStreamSynchronize(s));
    return ECGPU_SUCCESS;
}
static int patch_now(ecgpu_resident_state* st, const u64* offsets, const u64* data_off, const u8* data, u32 n);
// previous_epoch_participation <- current_epoch_participation <- zeros, on the device (nothing travels); both trees are rebuilt
// at the next root (2 x n / 32 hash64)
static int rotate_now(ecgpu_resident_state* st, u64 prev_start, u64 cur_start, u64 len) {
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    int rc = order_after_earlier_work(st, s);
    if (rc) return rc;
    ECG_HIP_CHECK(hipMemcpyAsync(st->d_ssz + prev_start, st->d_ssz + cur_start, len, hipMemcpyDeviceToDevice, s));
    ECG_HIP_CHECK(hipMemsetAsync(st->d_ssz + cur_start, 0, len, s));
    StatePlan plan;
    if (!build_state_plan(st->fork, st->h_fixed.data(), st->n_bytes, st->preset, plan, st->ext_roots(), nullptr)) {
        set_last_error(plan.error);
        return ECGPU_ERR_BAD_ARG;
    }
    for (u32 k = 0; k < plan.bigs.size() && k < TREE_MAX_FIELDS; k++)
        if ((plan.bigs[k].out_chunk == 15 || plan.bigs[k].out_chunk == 16) && plan.bigs[k].mix) st->trees.f[k].all_dirty = true;
    return record_change(st, s);
}
}  // namespace

// the resident state as csrc/state_fields.h's queue sees it
struct ResidentSink {
    ecgpu_resident_state* st;
    int fork() const { return st->fork; }
    int preset() const { return st->preset; }
    const u8* fixed() const { return st->h_fixed.data(); }
    u64 size() const { return st->n_bytes; }
    void fail(const char* m) { set_last_error(m); }
    // (the queue itself never touches HIP -- a queued write is ~30 ns of host work; the calling thread is bound to its device here,
    // where bytes actually move)
    int apply_patches(const u64* offsets, const u64* data_off, const u8* data, u32 n) {
        int rc = ensure_init();
        return rc ? rc : patch_now(st, offsets, data_off, data, n);
    }
    int apply_resize(u32 vi, const u8* data, u64 add_len, u64 keep, FieldResize mode) {
        int rc = ensure_init();
        if (rc) return rc;
        return resize_field(st, (int)vi, data, add_len, keep, mode == FIELD_APPEND ? RESIZE_APPEND : mode == FIELD_TRUNCATE ? RESIZE_TRUNCATE : RESIZE_REPLACE);
    }
    int apply_rotate(u64 prev_start, u64 cur_start, u64 len) {
        int rc = ensure_init();
        return rc ? rc : rotate_now(st, prev_start, cur_start, len);
    }
};
// everything queued through the field-addressed entries reaches the device before an operation addressed in bytes, a root,
// or a size query sees the state
static int flush_queue(ecgpu_resident_state* st) {
    if (!st->queue.pending()) return ECGPU_SUCCESS;
    ResidentSink sink{st};
    return st->queue.flush(sink);
}

int ecgpu_resident_state_append(ecgpu_resident_state_t* st, int field, const uint8_t* data, uint64_t n_bytes) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!st) return ECGPU_ERR_BAD_ARG;
    if ((rc = flush_queue(st))) return rc;
    return resize_field(st, field, data, n_bytes, 0, RESIZE_APPEND);
}

int ecgpu_resident_state_replace(ecgpu_resident_state_t* st, int field, const uint8_t* data, uint64_t n_bytes) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!st) return ECGPU_ERR_BAD_ARG;
    if ((rc = flush_queue(st))) return rc;
    return resize_field(st, field, data, n_bytes, 0, RESIZE_REPLACE);
}

int ecgpu_resident_state_truncate(ecgpu_resident_state_t* st, int field, uint64_t new_n_bytes) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!st) return ECGPU_ERR_BAD_ARG;
    if ((rc = flush_queue(st))) return rc;
    return resize_field(st, field, nullptr, 0, new_n_bytes, RESIZE_TRUNCATE);
}

// (the size the caller's program order implies: queued pushes count)
uint64_t ecgpu_resident_state_size(const ecgpu_resident_state_t* st) { return st ? st->n_bytes + st->queue.pushed_bytes() : 0; }

void ecgpu_resident_state_destroy(ecgpu_resident_state_t* st) {
    if (!st) return;
    (void)hipDeviceSynchronize();
    (void)hipFree(st->d_ssz);
    (void)hipFree(st->d_rootbuf);
    st->trees.release();
    if (st->patched) (void)hipEventDestroy(st->patched);
    if (st->rooted) (void)hipEventDestroy(st->rooted);
    delete st;
}

int ecgpu_resident_state_patch(ecgpu_resident_state_t* st, const uint64_t* offsets, const uint64_t* data_off, const uint8_t* data,
                               uint32_t n) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!st || (n && (!offsets || !data_off || !data))) return ECGPU_ERR_BAD_ARG;
    if ((rc = flush_queue(st))) return rc;  // byte offsets refer to the encoding as program order has left it
    return patch_now(st, offsets, data_off, data, n);
}

}  // extern "C"

namespace {
static int patch_now(ecgpu_resident_state* st, const u64* offsets, const u64* data_off, const u8* data, u32 n) {
    int rc = ECGPU_SUCCESS;
    if (!n) return ECGPU_SUCCESS;
    // (scratch vectors are the thread's: a slot's 4 096 patches are 200 KB of descriptors -- above malloc's mmap threshold, so a
    // fresh vector per call is an mmap, its page faults and an munmap)
    static thread_local std::vector<PatchDesc> descs;
    static thread_local std::vector<u64> pairs;
    descs.resize(n);
    pairs.clear();
    for (u32 i = 0; i < n; i++) {
        if (data_off[i + 1] < data_off[i]) return ECGPU_ERR_BAD_ARG;
        const u64 len = data_off[i + 1] - data_off[i];
        if (offsets[i] > st->n_bytes || len > st->n_bytes - offsets[i]) {
            set_last_error("patch outside the state encoding");
            return ECGPU_ERR_BAD_ARG;
        }
        descs[i] = {offsets[i], data_off[i], len};
    }
    // the variable-size lists keep their lengths: a patch must not rewrite the offset words of the fixed part
    const FixedLayout L = layout_for(STATE_PRESETS[st->preset], st->fork);
    u64 off_words[15] = {L.historical_roots_off, L.eth1_data_votes_off, L.validators_off, L.balances_off, L.prev_participation_off,
                         L.cur_participation_off, L.inactivity_scores_off, L.payload_header_off, L.historical_summaries_off, NO_FIELD,
                         L.pending_balance_deposits_off, L.pending_partial_withdrawals_off, L.pending_consolidations_off,
                         L.prev_attestations_off, L.cur_attestations_off};
    if (st->fork == FORK_PHASE0) {
        // the PendingAttestation lists are rooted where they are handed over (ecgpu_resident_state_replace): bytes patched
        // underneath would leave the two roots stale
        const u64 att0 = rd32(st->h_fixed.data() + L.prev_attestations_off);
        for (u32 i = 0; i < n; i++)
            if (descs[i].len && descs[i].dst_off + descs[i].len > att0) {
                set_last_error("a patch may not reach into the PendingAttestation lists: ecgpu_resident_state_replace");
                return ECGPU_ERR_BAD_ARG;
            }
    }
    // ... and the one offset word INSIDE the payload header (extra_data): the reference's deserializer rejects any other value
    if (L.payload_header_off != NO_FIELD) off_words[9] = rd32(st->h_fixed.data() + L.payload_header_off) + PAYLOAD_EXTRA_DATA_OFFSET_WORD;
    for (u32 i = 0; i < n; i++)
        for (u64 w : off_words)
            if (w != NO_FIELD && descs[i].dst_off < w + 4 && descs[i].dst_off + descs[i].len > w) {
                if (w >= st->h_fixed.size()) {  // the payload header's word: must stay == its fixed size
                    const u64 want = payload_header_fixed(st->fork);
                    bool same_w = true;
                    for (u64 b = (descs[i].dst_off > w ? descs[i].dst_off : w); b < w + 4 && b < descs[i].dst_off + descs[i].len; b++)
                        same_w = same_w && data[descs[i].src_off + (b - descs[i].dst_off)] == (u8)(want >> (8 * (b - w)));
                    if (!same_w) {
                        set_last_error("a patch may not change the extra_data offset of the payload header");
                        return ECGPU_ERR_BAD_ARG;
                    }
                    continue;
                }
                bool same = true;
                for (u64 b = (descs[i].dst_off > w ? descs[i].dst_off : w); b < w + 4 && b < descs[i].dst_off + descs[i].len; b++)
                    same = same && data[descs[i].src_off + (b - descs[i].dst_off)] == st->h_fixed[b];
                if (!same) {
                    set_last_error("a patch may not change an SSZ offset: reload the state when a list changes length");
                    return ECGPU_ERR_BAD_ARG;
                }
            }
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    Arena& ar = c->arena(s);
    ar.reset();
    const u64 total = data_off[n];
    // the level-0 entries of the cached trees these bytes belong to: marked on the device (state_tree.h MARK)
    for (u32 i = 0; i < n; i++)
        if (descs[i].len) st->trees.collect(descs[i].dst_off, descs[i].dst_off + descs[i].len, pairs);
    // ONE block travels: descriptors | marks | bytes.  Up to 1 MB it goes through a slot of the pinned upload ring -- one truly
    // asynchronous copy, and the caller's buffers are free when this returns (round 4: three pageable copies and a stream
    // synchronisation, 0.19 ms for a slot's 4 096 patches); a larger one (an epoch's rewrite of every balance) is copied
    // from where it lies, and the call waits for it.
    const size_t off_pairs = n * sizeof(PatchDesc), off_data = off_pairs + 8 * pairs.size(), block = off_data + total;
    rc = ar.reserve(block + 8192);
    if (rc) return st->trees.abandon_collected(), rc;
    u8* d_block = ar.take(block ? block : 1);
    if (!d_block) return st->trees.abandon_collected(), ECGPU_ERR_OOM;
    const bool ring = block <= (1u << 20);
    if (ring) {
        u8* h_block;
        hipEvent_t copied;
        rc = c->uploads.acquire(block, &h_block, &copied);
        if (rc) return st->trees.abandon_collected(), rc;
        std::memcpy(h_block, descs.data(), off_pairs);
        if (!pairs.empty()) std::memcpy(h_block + off_pairs, pairs.data(), 8 * pairs.size());
        if (total) std::memcpy(h_block + off_data, data, total);
        ECG_HIP_CHECK(hipMemcpyAsync(d_block, h_block, block, hipMemcpyHostToDevice, s));
        ECG_HIP_CHECK(hipEventRecord(copied, s));
    } else {
        ECG_HIP_CHECK(hipMemcpyAsync(d_block, descs.data(), off_pairs, hipMemcpyHostToDevice, s));
        if (!pairs.empty()) ECG_HIP_CHECK(hipMemcpyAsync(d_block + off_pairs, pairs.data(), 8 * pairs.size(), hipMemcpyHostToDevice, s));
        if (total) ECG_HIP_CHECK(hipMemcpyAsync(d_block + off_data, data, total, hipMemcpyHostToDevice, s));
    }
    if ((rc = order_after_earlier_work(st, s))) return rc;
    hipLaunchKernelGGL(k_apply_patches, dim3(n), dim3(64), 0, s, st->d_ssz, (const u8*)(d_block + off_data), (const PatchDesc*)d_block);
    ECG_HIP_CHECK(hipGetLastError());
    if (!pairs.empty()) {
        rc = st->trees.mark(s, (const u64*)(d_block + off_pairs), (u32)pairs.size());
        if (rc) return rc;
    }
    for (u32 i = 0; i < n; i++)  // host mirror of the fixed part
        for (u64 b = 0; b < descs[i].len; b++)
            if (descs[i].dst_off + b < st->h_fixed.size()) st->h_fixed[descs[i].dst_off + b] = data[descs[i].src_off + b];
    // a root may be asked for on another stream (ecgpu_resident_state_root_dev): it waits for this event, not the host
    if ((rc = record_change(st, s))) return rc;
    if (!ring) ECG_HIP_CHECK(hipStreamSynchronize(s));
    return ECGPU_SUCCESS;
}
}  // namespace

extern "C" {

// ---- field-addressed entries (round 6; csrc/state_fields.h) ------------------------------------------------------------------
#define ECG_FIELD_ENTRY(st)                       \
    int rc = ECGPU_SUCCESS;                       \
    (void)rc;                                     \
    if (!(st)) return ECGPU_ERR_BAD_ARG;          \
    ResidentSink sink { (st) }

int ecgpu_resident_state_patch_field(ecgpu_resident_state_t* st, uint32_t field, uint64_t offset_in_field, const uint8_t* data, uint64_t n_bytes) {
    ECG_FIELD_ENTRY(st);
    return st->queue.write(sink, field, offset_in_field, data, n_bytes);
}

int ecgpu_resident_state_patch_elements(ecgpu_resident_state_t* st, uint32_t field, uint64_t first_index, const uint8_t* data, uint64_t n_bytes) {
    ECG_FIELD_ENTRY(st);
    const FieldStatic f = field_static(st->fork, st->preset, field);
    if (!f.present || !f.elem || n_bytes % f.elem) {
        set_last_error("patch_elements: not a whole number of elements of this field");
        return ECGPU_ERR_BAD_ARG;
    }
    if (first_index > (~0ull) / f.elem) return ECGPU_ERR_BAD_ARG;
    return st->queue.write(sink, field, first_index * f.elem, data, n_bytes);
}

int ecgpu_resident_state_push(ecgpu_resident_state_t* st, uint32_t field, const uint8_t* data, uint64_t n_bytes) {
    ECG_FIELD_ENTRY(st);
    return st->queue.push(sink, field, data, n_bytes);
}

int ecgpu_resident_state_truncate_field(ecgpu_resident_state_t* st, uint32_t field, uint64_t new_n_bytes) {
    ECG_FIELD_ENTRY(st);
    return st->queue.truncate(sink, field, new_n_bytes);
}

int ecgpu_resident_state_set_field(ecgpu_resident_state_t* st, uint32_t field, const uint8_t* data, uint64_t n_bytes) {
    ECG_FIELD_ENTRY(st);
    return st->queue.set(sink, field, data, n_bytes);
}

int ecgpu_resident_state_add_validator(ecgpu_resident_state_t* st, const uint8_t validator121[121], uint64_t balance) {
    ECG_FIELD_ENTRY(st);
    if (!validator121) return ECGPU_ERR_BAD_ARG;
    return st->queue.add_validator(sink, validator121, balance);
}

int ecgpu_resident_state_rotate_participation(ecgpu_resident_state_t* st) {
    ECG_FIELD_ENTRY(st);
    return st->queue.rotate_participation(sink);
}

int ecgpu_resident_state_flush(ecgpu_resident_state_t* st) {
    ECG_FIELD_ENTRY(st);
    (void)sink;
    if (!st->queue.pending()) return ECGPU_SUCCESS;
    return flush_queue(st);
}

int64_t ecgpu_resident_state_field_size(ecgpu_resident_state_t* st, uint32_t field) {
    if (!st) return ECGPU_ERR_BAD_ARG;
    ResidentSink sink{st};
    FieldLoc loc;
    u64 seen = 0;
    if (!st->queue.locate(sink, field, loc, seen)) return ECGPU_ERR_BAD_ARG;
    return (int64_t)seen;
}

// Root of a resident state: (1) fields flagged for a rebuild are rebuilt level by level, (2) ONE climb launch re-hashes the
// dirty paths of every other cached field, (3) the fused tail runs one finishing job per cached field (<= 512 nodes -> zero
// ladder -> length mix-in), the small fields and the state container.
int ecgpu_resident_state_root_dev(ecgpu_resident_state_t* st, uint8_t* d_root, ecgpu_stream_t stream) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!st || !d_root) return ECGPU_ERR_BAD_ARG;
    if ((rc = flush_queue(st))) return rc;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(stream);
    if (st->patched && st->patch_stream != s) ECG_HIP_CHECK(hipStreamWaitEvent(s, st->patched, 0));
    u64 rebuilt = 0;
    rc = st->trees.update(s, st->d_ssz, &rebuilt);
    if (rc) return rc;
    rc = state_root_device(s, c, st->d_ssz, st->n_bytes, st->h_fixed.data(), st->preset, d_root, &st->trees, st->fork, st->ext_roots());
    c->last_hash64 += rebuilt;  // (the climbs' share is on the device: the host-pointer entry below adds it)
    if (rc) return rc;
    if (!st->rooted) ECG_HIP_CHECK(hipEventCreateWithFlags(&st->rooted, hipEventDisableTiming));
    ECG_HIP_CHECK(hipEventRecord(st->rooted, s));
    st->root_stream = s;
    return rc;
}

int ecgpu_resident_state_root(ecgpu_resident_state_t* st, uint8_t root[32]) {
    if (!st || !root) return ECGPU_ERR_BAD_ARG;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = flush_queue(st))) return rc;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    u8* d_root = st->d_rootbuf;
    if (st->patched && st->patch_stream != s) ECG_HIP_CHECK(hipStreamWaitEvent(s, st->patched, 0));
    u64 rebuilt = 0;
    rc = st->trees.update(s, st->d_ssz, &rebuilt);
    if (rc) return rc;
    rc = state_root_device(s, c, st->d_ssz, st->n_bytes, st->h_fixed.data(), st->preset, d_root, &st->trees, st->fork, st->ext_roots());
    if (rc) return rc;
    // root and the climbs' hash counter come back in one copy
    ECG_HIP_CHECK(hipMemcpyAsync(d_root + 32, st->trees.d_hashes(), 8, hipMemcpyDeviceToDevice, s));
    ECG_HIP_CHECK(hipMemsetAsync(st->trees.d_hashes(), 0, 8, s));
    rc = c->staging.reserve(64);
    if (rc) return rc;
    ECG_HIP_CHECK(hipMemcpyAsync(c->staging.p, d_root, 40, hipMemcpyDeviceToHost, s));
    ECG_HIP_CHECK(hipStreamSynchronize(s));
    std::memcpy(root, c->staging.p, 32);
    std::memcpy(&st->last_climb_hashes, c->staging.p + 32, 8);
    c->last_hash64 += rebuilt + st->last_climb_hashes;
    return ECGPU_SUCCESS;
}

uint64_t ecgpu_beacon_state_fixed_size(int fork, int preset) {
    if (preset < 0 || preset > 1 || fork < FORK_PHASE0 || fork > FORK_LAST) return 0;
    return layout_for(STATE_PRESETS[preset], fork).size;
}

// phase0 through the device entry ("missing" 4): the two PendingAttestation lists hold variable-size elements
// whose offset tables are part of the encoding, and the plan of such a list is built from those tables -- on the host.  They
// are the LAST two variable fields of a phase0 state (phase0/beacon_state.rs:80-81) and small (<= 4 096 attestations of a few
// hundred bytes), so the entry copies that tail of the encoding back ONCE (one synchronisation of the stream: the phase0
// form is not fully asynchronous, include/ecgpu.h says so), roots the two lists through the generic planner, and hands the two
// nodes to the device plan like the host entry does.
static int phase0_attestation_roots_dev(hipStream_t s, const u8* d_ssz, u64 n_bytes, const u8* h_fixed, int preset, u8 ext[64]) {
    if (preset < 0 || preset > 1) return ECGPU_ERR_BAD_ARG;
    const FixedLayout L = layout_for(STATE_PRESETS[preset], FORK_PHASE0);
    if (n_bytes < L.size) {
        set_last_error("state encoding shorter than its fixed part");
        return ECGPU_ERR_BAD_ARG;
    }
    const u64 a = rd32(h_fixed + L.prev_attestations_off), b = rd32(h_fixed + L.cur_attestations_off);
    if (a > b || b > n_bytes || a < L.size) {
        set_last_error("SSZ offsets not monotonic");
        return ECGPU_ERR_BAD_ARG;
    }
    std::vector<u8> tail(n_bytes - a ? n_bytes - a : 1);
    if (n_bytes - a) ECG_HIP_CHECK(hipMemcpyAsync(tail.data(), d_ssz + a, n_bytes - a, hipMemcpyDeviceToHost, s));
    ECG_HIP_CHECK(hipStreamSynchronize(s));
    int rc = pending_attestations_root(tail.data(), b - a, preset, ext);
    if (rc) return rc;
    return pending_attestations_root(tail.data() + (b - a), n_bytes - b, preset, ext + 32);
}

int ecgpu_htr_beacon_state_dev(int fork, const uint8_t* d_ssz, uint64_t n_bytes, const uint8_t* h_fixed, int preset, uint8_t* d_root,
                               ecgpu_stream_t stream) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!d_ssz || !h_fixed || !d_root || fork < FORK_PHASE0 || fork > FORK_LAST) return ECGPU_ERR_BAD_ARG;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(stream);
    if (fork == FORK_PHASE0) {
        u8 ext[64];
        if ((rc = phase0_attestation_roots_dev(s, d_ssz, n_bytes, h_fixed, preset, ext))) return rc;
        // (the generic planner used this thread's own stream and arena; the caller's stream is idle here)
        return state_root_device(s, c, d_ssz, n_bytes, h_fixed, preset, d_root, nullptr, fork, ext);
    }
    return state_root_device(s, c, d_ssz, n_bytes, h_fixed, preset, d_root, nullptr, fork);
}

int ecgpu_htr_beacon_state_dev_checked(int fork, const uint8_t* d_ssz, uint64_t n_bytes, const uint8_t* h_fixed, int preset,
                                       uint8_t* d_root, int32_t* d_status, ecgpu_stream_t stream) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!d_ssz || !h_fixed || !d_root || !d_status || fork < FORK_ALTAIR || fork > FORK_LAST) return ECGPU_ERR_BAD_ARG;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(stream);
    return state_root_device(s, c, d_ssz, n_bytes, h_fixed, preset, d_root, nullptr, fork, nullptr, nullptr, nullptr, (int*)d_status);
}

int ecgpu_beacon_state_shard_subroots_dev(int fork, const uint8_t* d_ssz, uint64_t n_bytes, const uint8_t* h_fixed, int preset,
                                          uint32_t rank, uint32_t world, uint8_t* d_subroots, uint8_t* d_field_roots,
                                          ecgpu_stream_t stream) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!d_ssz || !h_fixed || !d_subroots || fork < FORK_ALTAIR || fork > FORK_LAST || !world || world > 512 || rank >= world)
        return ECGPU_ERR_BAD_ARG;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(stream);
    return state_shard_subroots_device(s, c, d_ssz, n_bytes, h_fixed, preset, fork, rank, world, d_subroots, d_field_roots);
}

int ecgpu_htr_beacon_state_sharded_dev(int fork, const uint8_t* d_ssz, uint64_t n_bytes, const uint8_t* h_fixed, int preset,
                                       const uint8_t* d_all_subroots, uint32_t world, const uint8_t* d_field_roots, uint8_t* d_root,
                                       ecgpu_stream_t stream) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!d_ssz || !h_fixed || !d_all_subroots || !d_root || fork < FORK_ALTAIR || fork > FORK_LAST || !world || world > 512)
        return ECGPU_ERR_BAD_ARG;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(stream);
    const ShardTop top{d_all_subroots, world, d_field_roots};
    return state_root_device(s, c, d_ssz, n_bytes, h_fixed, preset, d_root, nullptr, fork, nullptr, nullptr, nullptr, nullptr, &top);
}

uint32_t ecgpu_beacon_state_shard_lists(void) { return N_SHARDED_LISTS; }

// List<PendingAttestation<MAX_VALIDATORS_PER_COMMITTEE>, MAX_ATTESTATIONS * SLOTS_PER_EPOCH> (phase0/operations.rs:45-52,
// phase0/presets/*.rs:84) for the generic planner
static int pending_attestations_root(const u8* ssz, u64 n_bytes, int preset, u8 root[32]) {
    static const uint32_t fields[] = {0, 1,           // Checkpoint: epoch, root
                                      0, 0, 1, 2, 2,  // AttestationData: slot, index, beacon_block_root, source, target
                                      4, 3, 0, 0};    // PendingAttestation: aggregation_bits, data, inclusion_delay, proposer_index
    const ecgpu_ssz_type types[7] = {{ECGPU_SSZ_UINT, 0, 8, 0, 0},          {ECGPU_SSZ_BYTEVECTOR, 0, 32, 0, 0},
                                     {ECGPU_SSZ_CONTAINER, 0, 0, 2, 0},     {ECGPU_SSZ_CONTAINER, 0, 0, 5, 2},
                                     {ECGPU_SSZ_BITLIST, 0, 2048, 0, 0},    {ECGPU_SSZ_CONTAINER, 0, 0, 4, 7},
                                     {ECGPU_SSZ_LIST, 5, preset == 0 ? 4096ull : 1024ull, 0, 0}};
    return ecgpu_htr_ssz(types, 7, fields, 11, 6, ssz, n_bytes, root);
}

static int beacon_state_host(int fork, const uint8_t* ssz, uint64_t n_bytes, int preset, uint8_t root[32], uint8_t* field_roots);

int ecgpu_htr_beacon_state(int fork, const uint8_t* ssz, uint64_t n_bytes, int preset, uint8_t root[32]) {
    return beacon_state_host(fork, ssz, n_bytes, preset, root, nullptr);
}

int ecgpu_beacon_state_field_roots(int fork, const uint8_t* ssz, uint64_t n_bytes, int preset, uint8_t* roots, uint32_t capacity,
                                   uint32_t* n_fields, uint8_t root[32]) {
    if (!roots || !n_fields || !root || fork < FORK_PHASE0 || fork > FORK_LAST || capacity < state_field_count(fork)) return ECGPU_ERR_BAD_ARG;
    u8 all[32 * STATE_MAX_FIELD_CHUNKS];
    int rc = beacon_state_host(fork, ssz, n_bytes, preset, root, all);
    if (rc) return rc;
    *n_fields = state_field_count(fork);
    std::memcpy(roots, all, 32ull * *n_fields);
    return ECGPU_SUCCESS;
}

static int beacon_state_host(int fork, const uint8_t* ssz, uint64_t n_bytes, int preset, uint8_t root[32], uint8_t* field_roots) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!ssz || !root) return ECGPU_ERR_BAD_ARG;
    if (preset < 0 || preset > 1 || fork < FORK_PHASE0 || fork > FORK_LAST || n_bytes < layout_for(STATE_PRESETS[preset], fork).size) {
        set_last_error("bad fork / preset or truncated state");
        return ECGPU_ERR_BAD_ARG;
    }
    const FixedLayout L = layout_for(STATE_PRESETS[preset], fork);
    u8 ext[64];
    if (fork == FORK_PHASE0) {
        // the two lists of variable-size elements go through the generic planner first (their encodings are on the host here)
        const u64 a = rd32(ssz + L.prev_attestations_off), b = rd32(ssz + L.cur_attestations_off);
        if (a > b || b > n_bytes) {
            set_last_error("SSZ offsets not monotonic");
            return ECGPU_ERR_BAD_ARG;
        }
        if ((rc = pending_attestations_root(ssz + a, b - a, preset, ext))) return rc;
        if ((rc = pending_attestations_root(ssz + b, n_bytes - b, preset, ext + 32))) return rc;
    }
    const u8* h_payload = nullptr;
    if (fork >= FORK_BELLATRIX) {
        const u64 h = rd32(ssz + L.payload_header_off);
        if (h > n_bytes || n_bytes - h < payload_header_fixed(fork)) {
            set_last_error("payload header outside the encoding");
            return ECGPU_ERR_BAD_ARG;
        }
        h_payload = ssz + h;
    }
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    // the encoding lives in its own allocation (the arena is rebuilt by the driver)
    struct StateBuf {
        u8* p = nullptr;
        size_t cap = 0;
    };
    static thread_local std::map<int, StateBuf> bufs;  // per (host thread, device)
    u8*& d_state = bufs[current_device()].p;
    size_t& d_state_cap = bufs[current_device()].cap;
    if (n_bytes + 64 > d_state_cap) {
        if (d_state) {
            ECG_HIP_CHECK(hipStreamSynchronize(s));
            ECG_HIP_CHECK(hipFree(d_state));
            d_state = nullptr;
            d_state_cap = 0;
        }
        ECG_HIP_CHECK(hipMalloc((void**)&d_state, n_bytes + 4096 + (n_bytes >> 3)));  // + the root and up to 64 field roots
        d_state_cap = n_bytes + 64 + (n_bytes >> 3);
    }
    ECG_HIP_CHECK(hipMemcpyAsync(d_state, ssz, n_bytes, hipMemcpyHostToDevice, s));
    u8* d_root = d_state + ((n_bytes + 31) / 32) * 32;
    rc = state_root_device(s, c, d_state, n_bytes, ssz, preset, d_root, nullptr, fork, fork == FORK_PHASE0 ? ext : nullptr, h_payload,
                           field_roots ? d_root + 32 : nullptr);
    if (rc) return rc;
    if (field_roots) ECG_HIP_CHECK(hipMemcpyAsync(field_roots, d_root + 32, 32 * state_field_chunks(fork), hipMemcpyDeviceToHost, s));
    ECG_HIP_CHECK(hipMemcpyAsync(root, d_root, 32, hipMemcpyDeviceToHost, s));
    ECG_HIP_CHECK(hipStreamSynchronize(s));
    return ECGPU_SUCCESS;
}

uint64_t ecgpu_beacon_state_deneb_fixed_size(int preset) {
    if (preset < 0 || preset > 1) return 0;
    return layout_for(STATE_PRESETS[preset]).size;
}

int ecgpu_htr_beacon_state_deneb_dev(const uint8_t* d_ssz, uint64_t n_bytes, const uint8_t* h_fixed, int preset,
                                     uint8_t* d_root, ecgpu_stream_t stream) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!d_ssz || !h_fixed || !d_root) return ECGPU_ERR_BAD_ARG;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(stream);
    return state_root_device(s, c, d_ssz, n_bytes, h_fixed, preset, d_root);
}

int ecgpu_htr_beacon_state_deneb(const uint8_t* ssz, uint64_t n_bytes, int preset, uint8_t root[32]) {
    return ecgpu_htr_beacon_state(FORK_DENEB, ssz, n_bytes, preset, root);
}

}  // extern "C"
