// Internal host-side interface of the Merkle engine (merkle.hip) used by the ABI entry points
// and by the BeaconState driver (state_deneb.hip).
#pragma once
#include "merkle.h"
#include "state_plan.h"
#include "runtime.h"

namespace ecg {

inline u64 leaf_record_bytes(LeafKind k) {
    switch (k) {
        case LEAF_VALIDATORS: return 121;
        case LEAF_BYTES48: return 48;
        case LEAF_PAIR64: return 64;
        case LEAF_ETH1DATA: return 72;
        case LEAF_U64X2: return 16;
        case LEAF_U64X3: return 24;
        default: return 32;
    }
}

// device workspace (bytes) needed by merkleize_device for n level-0 nodes
size_t merkle_ws_bytes(u64 n0);

// Enqueue on `s`: root of the tree over `n0` level-0 nodes produced by `kind` from d_in
// (in_bytes long), climbed to `depth`, optionally mixed with `mix_len`; 32-byte result to d_out.
// `ws` must provide merkle_ws_bytes(n0).  Adds the number of hash64 performed to *hash_count.
// With `deferred` set, the finishing job (<= 512 nodes -> root) is NOT launched: it is returned with offsets
// relative to `job_base` so that the caller can batch it with others (launch_tree_jobs); `background`
// selects the fewer-launches pass schedule (state_plan.h); `after_wide_passes` is recorded on `s` once the wide
// (chip-filling) passes are enqueued, i.e. where the tree's latency-bound tail begins.
// With `deferred_tile` set as well, the tile stage is not launched either: its descriptor (first_wg left 0) and workgroup
// count are returned (*deferred_tile_wgs = 0: the tree has no tile stage) for the fused tail of a BeaconState root.
// `phase`: MERKLEIZE_DESCRIBE fills the deferred descriptors and the hash count and launches nothing (the caller needs them
// before it enqueues anything: state_deneb.hip uploads its whole plan in front of the passes); MERKLEIZE_LAUNCH enqueues
// the passes of the same tree and touches neither the descriptors nor the count.
enum { MERKLEIZE_ALL = 0, MERKLEIZE_DESCRIBE = 1, MERKLEIZE_LAUNCH = 2 };
int merkleize_device(hipStream_t s, LeafKind kind, const u8* d_in, u64 in_bytes, u64 n0, u32 depth,
                     bool mix, u64 mix_len, u8* d_out, u8* ws, u64* hash_count, TreeJob* deferred = nullptr,
                     const u8* job_base = nullptr, bool background = false, hipEvent_t after_wide_passes = nullptr,
                     TileDesc* deferred_tile = nullptr, u32* deferred_tile_wgs = nullptr, int phase = MERKLEIZE_ALL);

// Batched small trees: jobs live in device memory at d_jobs.
int launch_tree_jobs(hipStream_t s, const TreeJob* d_jobs, u32 n_jobs, u8* d_buf);

// Batched tile stage: descriptors (merkle.h TileDesc) in device memory, n_wg = total workgroups.
int launch_tiles(hipStream_t s, const TileDesc* d_descs, u32 n_desc, u32 n_wg);

// chk_off != ~0: the gather kernel also compares the little-endian u32 at byte chk_off of the source with chk_expect and sets
// *d_flag = 1 when they differ (a check the host cannot make when it never sees the bytes: ecgpu_htr_beacon_state_dev)
int launch_gather(hipStream_t s, const u8* d_src, u64 src_total, const GatherDesc* d_desc, u32 n, u8* d_dst, u64 chk_off = ~0ull,
                  u32 chk_expect = 0, u32* d_flag = nullptr);

// ---- the fused tail of a BeaconState root (state_deneb.hip) -----------------------------------------------------------------
// After the wide passes, everything that is left of a state root is latency: per field a tile stage (1024 nodes -> 1 per
// workgroup), a finishing job (<= 512 nodes -> root, zero ladder, length mix-in), then the nested containers and the state
// container.  Round 2 ran that as ~6 dependent launches on two streams joined by events.  k_state_tail is ONE launch chained by
// ARRIVAL TICKETS: every tile workgroup takes a ticket of its field when its node is stored, and the LAST one runs the field's
// finishing job; a finished unit whose root is an input of a nested container takes a ticket of THAT container, and the last
// one runs it; every unit that feeds the state container directly, and every nested container, takes a ticket of the state
// container, and the last one computes it and writes the root.  No workgroup ever waits for another (no spinning, nothing to
// deadlock): whoever arrives last carries on.
// The chunks that are COPIED out of the encoding (basic fields, the members of the small containers: GatherDesc) are fetched
// by the unit that hashes them, right before it does -- there is no gather launch in front of the tail, and the check of the
// payload header's extra_data offset (the _dev entry, whose host never sees the bytes) is made by the unit that writes the root.
constexpr u32 TAIL_NONE = 0xffffffffu;
struct TailField {
    TileDesc tile;
    TreeJob job;
    u32 n_tiles;
    u32 feeds;  // index of the nested container this field's root is an input of, or TAIL_NONE
    u32 prio;   // 1: the field on the critical path (its chain waves get issue priority)
    u32 pad_;
};
constexpr u32 TAIL_MAX_FIELDS = 24, TAIL_MAX_JOBS0 = 64, TAIL_MAX_JOBS1 = 8;
struct TailPlan {
    TailField fields[TAIL_MAX_FIELDS];  // fields with a tile stage, in workgroup order
    TreeJob jobs0[TAIL_MAX_JOBS0];      // units without one: leaf containers, finishing jobs of short or empty fields
    u32 jobs0_feeds[TAIL_MAX_JOBS0];
    TreeJob jobs1[TAIL_MAX_JOBS1];      // nested containers
    u32 jobs1_deps[TAIL_MAX_JOBS1];     // units feeding each (>= 1)
    TreeJob job2;                       // the state container
    u32 n_fields, n_tile_wgs, n_jobs0, n_jobs1, final_parties;
    u32 n_froots;                       // chunk slots of the state container: 32, or 64 from electra on (37 fields)
    u64 root_off, froots_off;           // byte offsets in the job buffer: the state root, the n_froots field-root chunks
    u8* d_root;
    u8* d_field_roots;                  // may be null
    u32* counters;                      // [n_fields] tile tickets, [n_jobs1] nested containers, [1] the state container; zero before the launch
    const u8* src;                      // the SSZ encoding and its length: source of the gathered chunks
    u64 src_total;
    const GatherDesc* gathers;
    u32 n_gathers, chk_expect;
    u64 small_off;                      // byte offset of the small-chunk buffer (GatherDesc::dst_chunk counts from there)
    u64 small_end;                      // ... and of its end: only jobs whose inputs lie in [small_off, small_end) have gathered chunks
    u64 chk_off;                        // != ~0: the little-endian u32 at src + chk_off must equal chk_expect, else the root is 32 x 0xFF
    const u8* ext_src;                  // gathers with src_sel == 1 read here (ext_total bytes): nodes computed elsewhere
    u64 ext_total;
    const u8* ext2_src;                 // gathers with src_sel == 2: the field-root block phase A of a sharded root left (64 x 32 bytes)
    int* d_status;                      // may be null: 0 / ECGPU_ERR_BAD_ARG next to the root, see ecgpu_htr_beacon_state_dev_checked
};
int launch_state_tail(hipStream_t s, const TailPlan* d_plan, u32 n_wgs, u8* d_buf);

const ZeroTable* device_zero_table();

}  // namespace ecg
