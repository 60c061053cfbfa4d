// Host side of the resident field trees (state_tree.h): allocation and geometry per field, patch -> (field, entry) marks,
// rebuild / climb launches, and the finishing job each tree hands to the fused tail of a state root.
#pragma once
#include <vector>

#include "merkle_driver.h"
#include "state_tree.h"

namespace ecg {

struct FieldTree {
    TreeGeom g{};        // device pointers + geometry (g.src is filled per launch from src_off)
    u8* block = nullptr;  // the one allocation behind lvl0 / nodes / cnt / flag0
    u64 src_off = 0;     // byte offset of the field in the encoding
    u32 out_chunk = 0, depth = 0;
    bool mix = false;
    u64 mix_len = 0;
    bool live = false;       // cached (>= TREE_MIN_ENTRIES level-0 entries)
    bool all_dirty = true;   // rebuilt from scratch at the next root
    u64 bound = 0;           // dirty-list entries this field may hold (pairs handed to mark since the last root)
    u64 region_bits[8] = {};  // regions with a mark since the last root (<= 512): the climb launch gets one workgroup per set bit
    u64 share() const { return (1ull << g.H) / 8 > 64 ? (1ull << g.H) / 8 : 64; }  // beyond this many marks a rebuild is cheaper: all_dirty
};

struct ResidentTrees {
    FieldTree f[TREE_MAX_FIELDS];  // slot = position of the field in StatePlan::bigs (stable for a fork)
    u32 n_slots = 0;
    u32* d_active = nullptr;       // active regions: (slot << 16) | region, TREE_ACTIVE_CAP entries
    u32* d_count = nullptr;        // [0] active regions listed; [2..3] u64: hash64 performed by climbs since it was last cleared
    u64 bound_total = 0;           // dirty entries handed to mark since the last root (>= active regions)
    u32 active_regions = 0;        // regions marked since the last root, counted on the host: the device list has as many entries
    u32 max_T = 0;                 // largest region height among the live fields: LDS of the climb launch

    // after create and after every length change: field offsets / counts from the plan; a field whose height changed (or that
    // is new) is reallocated and rebuilt at the next root
    int sync_geometry(const StatePlan& plan);
    void release();
    // entries of cached fields that the byte range [lo, hi) of the encoding touches -> pairs (slot << 56 | entry); a field
    // whose share of the list would overflow is flagged all_dirty instead
    void collect(u64 lo, u64 hi, std::vector<u64>& pairs);
    void collect_entries(u32 slot, u64 first, u64 last, std::vector<u64>& pairs);
    // an error between collect() and mark(): the host's accounting describes marks that never reached the device (advisor, round 5).
    // Nothing was changed on the device either, so no root can be wrong -- the fields with pending accounting are simply rebuilt
    void abandon_collected() {
        for (u32 s = 0; s < n_slots; s++)
            if (f[s].live && f[s].bound) f[s].all_dirty = true;
    }
    // enqueue the marks (pairs already on the device)
    int mark(hipStream_t s, const u64* d_pairs, u32 n);
    // enqueue rebuilds of all_dirty fields and ONE climb launch over the dirty list; *hashes += the host-known part (rebuilds)
    int update(hipStream_t s, const u8* d_ssz, u64* hashes);
    // the unit the fused tail runs for slot: <= 512 nodes of level T -> zero ladder -> mix-in; offsets relative to `base`
    TreeJob job(u32 slot, const u8* base, u64 out_off) const;
    u64 job_hashes(u32 slot) const;
    unsigned long long* d_hashes() const { return reinterpret_cast<unsigned long long*>(d_count + 2); }
};

}  // namespace ecg
