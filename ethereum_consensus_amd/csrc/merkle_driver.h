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
int merkleize_device(hipStream_t s, LeafKind kind, const u8* d_in, u64 in_bytes, u64 n0, u32 depth,
                     bool mix, u64 mix_len, u8* d_out, u8* ws, u64* hash_count, TreeJob* deferred = nullptr,
                     const u8* job_base = nullptr, bool background = false, hipEvent_t after_wide_passes = nullptr);

// Batched small trees: jobs live in device memory at d_jobs.
int launch_tree_jobs(hipStream_t s, const TreeJob* d_jobs, u32 n_jobs, u8* d_buf);

// Batched tile stage: descriptors (merkle.h TileDesc) in device memory, n_wg = total workgroups.
int launch_tiles(hipStream_t s, const TileDesc* d_descs, u32 n_desc, u32 n_wg);

int launch_gather(hipStream_t s, const u8* d_src, u64 src_total, const GatherDesc* d_desc, u32 n, u8* d_dst);

const ZeroTable* device_zero_table();

}  // namespace ecg
