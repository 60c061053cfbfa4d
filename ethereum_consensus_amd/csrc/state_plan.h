// Host-side plan of hash_tree_root(BeaconState) for the deneb fork: pure offset arithmetic over
// the state's SSZ encoding, no hashing and no HIP calls (so tests/hostsim can execute the very
// same plan on the CPU lane simulator).
//
// Type tree: /root/reference/ethereum-consensus/src/deneb/beacon_state.rs:13-64 (28 fields),
// sub-containers phase0/beacon_state.rs:15-22 (Fork), phase0/beacon_block.rs:83-91 (header),
// phase0/operations.rs:13-17,66-71 (Checkpoint, Eth1Data), altair/sync.rs:17-22 (SyncCommittee),
// deneb/execution_payload.rs:48-76 (ExecutionPayloadHeader); limits from
// phase0/presets/{mainnet,minimal}.rs, altair/presets/mainnet.rs:19.
#pragma once
#include <string>
#include <vector>

#include "merkle.h"

namespace ecg {

enum LeafKind : int {
    LEAF_CHUNKS = 0,      // packed bytes -> 32-byte chunks
    LEAF_NODES = 1,       // 32-byte nodes in a workspace
    LEAF_VALIDATORS = 2,  // 121-byte Validator records -> htr(Validator)
    LEAF_BYTES48 = 3,     // 48-byte records -> htr(ByteVector<48>)
    LEAF_PAIR64 = 4,      // 64-byte records -> hash64 (two-field containers of roots)
    LEAF_ETH1DATA = 5,    // 72-byte Eth1Data records -> htr(Eth1Data)
    LEAF_U64X2 = 6,       // 16-byte records of two uint64 -> hash64 of their chunks (electra PendingBalanceDeposit, PendingConsolidation)
    LEAF_U64X3 = 7,       // 24-byte records of three uint64 -> htr of the 3-field container (electra PendingPartialWithdrawal)
};

inline u32 ceil_log2_u64(u64 x) {
    u32 d = 0;
    while (d < 64 && (1ull << d) < x) d++;
    return d;
}

// ---- pass schedule of one merkleize ----------------------------------------------------------
// Height of the subtree one lane reduces in a pass over `n` inputs: keep >= 2^18 lanes in flight
// (4 waves per SIMD on 256 CUs) while the tree is wide, go level by level once it is narrow
// (a narrow level is latency-bound by one hash64 per lane whatever D is).  `background` trees run
// underneath a bigger one on another stream: their latency is hidden, so they trade it for fewer
// launches (>= 2^14 lanes per pass).
inline int choose_pass_height(u64 n, bool background = false) {
    int lg = 63 - __builtin_clzll(n | 1);
    int d = lg - (background ? 14 : 18);
    if (d < 1) d = 1;
    if (d > 6) d = 6;
    return d;
}

struct PassStep {
    int D;         // subtree height per lane
    u64 n_in;      // level-0 nodes (first pass) or nodes of the previous pass
    u64 n_out;
    u32 level_in;  // absolute level of the inputs (0 for the first pass)
    bool first;    // runs the leaf functor of the field
};

struct MerkleSchedule {
    std::vector<PassStep> passes;
    // optional tile stage (merkle.h TileDesc) between the passes and the finishing job
    bool tile = false;
    bool tile_first = false;   // the tile stage runs the field's leaf functor (no pass before it)
    u64 tile_n_in = 0;
    u32 tile_level_in = 0;
    u32 job_n, job_level;  // finishing job: job_n nodes at job_level -> depth (+ mix-in)
    u64 hashes;            // hash64 of the tree (schedule-independent): leaf functors + every non-virtual node
};

inline u64 leaf_hash_cost(LeafKind k) {
    switch (k) {
        case LEAF_VALIDATORS: return 8;
        case LEAF_BYTES48:
        case LEAF_PAIR64:
        case LEAF_U64X2: return 1;
        case LEAF_ETH1DATA:
        case LEAF_U64X3: return 3;
        default: return 0;
    }
}

inline u64 tree_hash_count(LeafKind kind, u64 n0, u32 depth, bool mix) {
    u64 h = n0 * leaf_hash_cost(kind);
    u64 c = n0;
    if (n0)
        for (u32 l = 0; l < depth; l++) {
            c = (c + 1) / 2;
            h += c;
        }
    return h + (mix ? 1 : 0);
}

// Wide levels: depth-first passes (>= 2^18 lanes each) until at most TILE_MAX_IN nodes are left; then ONE tile
// stage (1024 nodes per workgroup, 10 levels) and the finishing job.  Validator records always get a pass of their
// own (8 hash64 per record, 92 VGPRs: it is the kernel the roofline is quoted on).
inline MerkleSchedule schedule_merkleize(LeafKind kind, u64 n0, u32 depth, bool mix, bool background = false) {
    MerkleSchedule sc;
    sc.hashes = tree_hash_count(kind, n0, depth, mix);
    u64 n = n0;
    u32 level = 0;
    bool first = (kind != LEAF_NODES);  // a leaf functor still has to run
    auto push_pass = [&](int D) {
        const u64 n_out = (n + (1ull << D) - 1) >> D;
        sc.passes.push_back({D, n, n_out, level, first});
        level += (u32)D;
        n = n_out;
        first = false;
    };
    // (Round 3 tried a throughput pre-pass for background fields wider than 2^15 nodes, on an auxiliary stream underneath the
    // validator pass: the fused tail got shorter, 0.46 -> 0.39 ms, and the validator pass longer, 0.62 -> 0.68 ms -- 1.14 ms
    // per root against 1.09: profiles/r03j_merkle_prepass_kernel_stats.txt.  Not kept.)
    while (n > 0 && ((first && kind == LEAF_VALIDATORS) || n > TILE_MAX_IN)) {
        int D = choose_pass_height(n, background);
        // (round 5) the registry's leaf pass is the LDS-staged one (merkle.hip k_merkle_pass<2, ValidatorLeaves>: four records per
        // lane) however long the registry: taller lanes would hold 8 or 16 roots across the calls
        if (first && kind == LEAF_VALIDATORS && D > 2) D = 2;
        if ((u32)D > depth - level) D = (int)(depth - level);
        if (!first && D == 0) break;
        push_pass(D);
    }
    if (n > 0 && (first || n > TREEJOB_MAX_NODES)) {
        if (depth - level >= TILE_D) {
            sc.tile = true;
            sc.tile_first = first;
            sc.tile_n_in = n;
            sc.tile_level_in = level;
            const u32 up = depth - level < TILE_LEVELS ? depth - level : TILE_LEVELS;
            n = up == TILE_LEVELS ? (n + TILE_NODES - 1) / TILE_NODES : 1;
            level += up;
            first = false;
        } else {
            push_pass((int)(depth - level));  // trees of height 0 or 1 behind a leaf functor
        }
    }
    sc.job_n = (u32)n;
    sc.job_level = level;
    return sc;
}

// Gather byte ranges of the encoding into zero-padded 32-byte chunks.
struct GatherDesc {
    u64 src_off;
    u32 n_bytes;    // <= 32
    u32 dst_chunk;  // chunk index in the small-chunk buffer
    u32 last_and;   // 0: copy as is; else AND mask for the last copied byte (Bitlist delimiter removal)
    u32 src_sel;    // 0: src_off counts from the SSZ encoding; 1: from the caller's node buffer (sharded state roots: the sub-roots
                    // of the registry-sized lists, all-gathered over the ranks -- merkle_driver.h TailPlan::ext_src); 2: from the
                    // field-root block phase A of a sharded root left behind (TailPlan::ext2_src)
};

// A big field: reduced by the pass kernels straight from the encoding.
struct BigField {
    LeafKind kind;
    u64 src, bytes, n0;
    u32 depth;
    bool mix;
    u64 mix_len;
    u32 out_chunk;
};

struct StatePlan {
    std::vector<GatherDesc> gathers;
    std::vector<TreeJob> jobs[3];  // dependency levels: leaf containers, nested containers, the state
    std::vector<BigField> bigs;
    u32 n_small_chunks = 0;
    u32 root_chunk = 0;
    u64 small_hashes = 0;  // hash64 performed by the jobs
    struct ExtChunk {
        u32 dst_chunk, src_off;  // 32 bytes of the caller's ext_roots buffer -> this chunk (phase0 attestation-list roots)
    };
    std::vector<ExtChunk> ext_chunks;
    u64 payload_header_off = ~0ull;  // byte offset of the payload header in the encoding (bellatrix+)
    std::string error;
};

struct Preset {
    u64 slots_per_historical_root, historical_roots_limit, eth1_data_votes_bound, validator_registry_limit,
        epochs_per_historical_vector, epochs_per_slashings_vector, sync_committee_size;
    // electra (electra/presets/{mainnet,minimal}.rs:10-12)
    u64 pending_balance_deposits_limit, pending_partial_withdrawals_limit, pending_consolidations_limit;
};
static const Preset STATE_PRESETS[2] = {
    {8192, 1ull << 24, 2048, 1ull << 40, 65536, 8192, 512, 1ull << 27, 1ull << 27, 1ull << 18},  // mainnet
    {64, 1ull << 24, 32, 1ull << 40, 64, 64, 32, 1ull << 27, 1ull << 6, 1ull << 6},              // minimal
};

// The forks whose BeaconState this plan knows: phase0/beacon_state.rs:50-88 (21 fields),
// altair/beacon_state.rs:13-55 (24), bellatrix/beacon_state.rs:13-58 (25: + latest_execution_payload_header),
// capella/beacon_state.rs:13-64 (28: + withdrawal indices, historical_summaries), deneb/beacon_state.rs:13-64 (28).
// Fields 0..14 are the same in every fork; altair replaced phase0's two PendingAttestation lists (fields 15, 16) by the
// participation-flag lists and appended inactivity_scores and the sync committees; the payload header grew from 14 fields
// (bellatrix/execution_payload.rs:58-81) to 15 (capella: + withdrawals_root) to 17 (deneb: + blob_gas_used, excess_blob_gas).
// electra (electra/beacon_state.rs:73-145; round 4, below SURVEY 8f): 37 fields -- deneb's 28, six uint64 and three lists of
// small fixed-size containers -- in a 64-leaf container; the payload header has 19 fields (electra/execution_payload.rs:54-84).
enum StateFork : int { FORK_PHASE0 = 0, FORK_ALTAIR = 1, FORK_BELLATRIX = 2, FORK_CAPELLA = 3, FORK_DENEB = 4, FORK_ELECTRA = 5 };
constexpr int FORK_LAST = FORK_ELECTRA;
constexpr u32 STATE_MAX_FIELD_CHUNKS = 64;  // chunks 0 .. 63 of the small buffer: the roots of the state's fields
constexpr u64 NO_FIELD = ~0ull;

inline u32 state_field_count(int fork) {
    return fork == FORK_PHASE0 ? 21u : fork == FORK_ALTAIR ? 24u : fork == FORK_BELLATRIX ? 25u : fork == FORK_ELECTRA ? 37u : 28u;
}
// leaves of the fork's container tree: 32, or 64 from electra on
inline u32 state_field_chunks(int fork) { return fork == FORK_ELECTRA ? 64u : 32u; }
// fixed part of the fork's ExecutionPayloadHeader (the offset word of extra_data sits at byte 436 and must hold this value)
inline u64 payload_header_fixed(int fork) { return fork == FORK_BELLATRIX ? 536 : fork == FORK_CAPELLA ? 568 : fork == FORK_ELECTRA ? 648 : 584; }

// byte offsets of the fields inside the fixed-size part of the encoding (NO_FIELD: the fork has no such field)
struct FixedLayout {
    u64 genesis_time, genesis_validators_root, slot, fork, latest_block_header, block_roots, state_roots,
        historical_roots_off, eth1_data, eth1_data_votes_off, eth1_deposit_index, validators_off, balances_off,
        randao_mixes, slashings, prev_participation_off, cur_participation_off, justification_bits,
        prev_justified, cur_justified, finalized, inactivity_scores_off, current_sync_committee,
        next_sync_committee, payload_header_off, next_withdrawal_index, next_withdrawal_validator_index,
        historical_summaries_off, prev_attestations_off, cur_attestations_off,
        deposit_receipts_start_index, pending_balance_deposits_off, pending_partial_withdrawals_off, pending_consolidations_off, size;
};

inline FixedLayout layout_for(const Preset& p, int fork = FORK_DENEB) {
    FixedLayout L;
    u64 o = 0;
    auto take = [&](u64 n) { u64 r = o; o += n; return r; };
    auto take_if = [&](bool have, u64 n) { return have ? take(n) : NO_FIELD; };
    const bool altair = fork >= FORK_ALTAIR, bellatrix = fork >= FORK_BELLATRIX, capella = fork >= FORK_CAPELLA, electra = fork >= FORK_ELECTRA;
    L.genesis_time = take(8);
    L.genesis_validators_root = take(32);
    L.slot = take(8);
    L.fork = take(16);
    L.latest_block_header = take(112);
    L.block_roots = take(32 * p.slots_per_historical_root);
    L.state_roots = take(32 * p.slots_per_historical_root);
    L.historical_roots_off = take(4);
    L.eth1_data = take(72);
    L.eth1_data_votes_off = take(4);
    L.eth1_deposit_index = take(8);
    L.validators_off = take(4);
    L.balances_off = take(4);
    L.randao_mixes = take(32 * p.epochs_per_historical_vector);
    L.slashings = take(8 * p.epochs_per_slashings_vector);
    L.prev_attestations_off = take_if(!altair, 4);
    L.cur_attestations_off = take_if(!altair, 4);
    L.prev_participation_off = take_if(altair, 4);
    L.cur_participation_off = take_if(altair, 4);
    L.justification_bits = take(1);
    L.prev_justified = take(40);
    L.cur_justified = take(40);
    L.finalized = take(40);
    L.inactivity_scores_off = take_if(altair, 4);
    L.current_sync_committee = take_if(altair, 48 * p.sync_committee_size + 48);
    L.next_sync_committee = take_if(altair, 48 * p.sync_committee_size + 48);
    L.payload_header_off = take_if(bellatrix, 4);
    L.next_withdrawal_index = take_if(capella, 8);
    L.next_withdrawal_validator_index = take_if(capella, 8);
    L.historical_summaries_off = take_if(capella, 4);
    // electra: deposit_receipts_start_index, deposit_balance_to_consume, exit_balance_to_consume, earliest_exit_epoch,
    // consolidation_balance_to_consume, earliest_consolidation_epoch (six uint64), then three list offsets
    L.deposit_receipts_start_index = take_if(electra, 48);
    L.pending_balance_deposits_off = take_if(electra, 4);
    L.pending_partial_withdrawals_off = take_if(electra, 4);
    L.pending_consolidations_off = take_if(electra, 4);
    L.size = o;
    return L;
}

inline u32 rd32(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }

constexpr u64 PAYLOAD_HEADER_FIXED = 584;  // deneb ExecutionPayloadHeader fixed part (extra_data offset at 436)
constexpr u64 PAYLOAD_EXTRA_DATA_OFFSET_WORD = 436;

struct Builder {
    std::vector<GatherDesc> gathers;
    std::vector<TreeJob> jobs[3];
    u32 next_chunk = STATE_MAX_FIELD_CHUNKS;  // chunks 0 .. 63 = the state container's field roots (37 of them in electra)
    u64 hashes = 0;
    u32 alloc(u32 n) { u32 r = next_chunk; next_chunk += n; return r; }
    void gather(u64 src, u32 nbytes, u32 dst_chunk) { gathers.push_back({src, nbytes, dst_chunk, 0u, 0u}); }
    void job(int lvl, u32 in_chunk, u32 n, u32 depth, u32 out_chunk, bool mix = false, u64 mix_len = 0) {
        TreeJob j;
        j.in_off = 32ull * in_chunk;
        j.out_off = 32ull * out_chunk;
        j.n = n;
        j.level = 0;
        j.depth = depth;
        j.mix = mix ? 1 : 0;
        j.mix_len = mix_len;
        jobs[lvl].push_back(j);
        u64 c = n, l = 0;
        while (c > 1) { c = (c + 1) / 2; hashes += c; l++; }
        if (n) hashes += depth - l;
        if (mix) hashes++;
    }
};

// Returns false (plan.error set) when the encoding is malformed.  `h_fixed`: at least the fixed part of the encoding.
// phase0: the roots of the two PendingAttestation lists (variable-size elements: the generic planner of ssz_plan.h computes
// them) are supplied by the caller as two 32-byte nodes in `ext_roots`; `h_payload_fixed` (optional, bellatrix+): the first
// 440 bytes of the payload header on the host, to check its extra_data offset word like the reference's deserializer does.
inline bool build_state_plan(int fork, const u8* h_fixed, u64 n_bytes, int preset, StatePlan& plan, const u8* ext_roots = nullptr,
                             const u8* h_payload_fixed = nullptr) {
    auto fail = [&](const char* m) { plan.error = m; return false; };
    if (preset < 0 || preset > 1) return fail("bad preset");
    if (fork < FORK_PHASE0 || fork > FORK_LAST) return fail("unknown fork");
    const Preset& P = STATE_PRESETS[preset];
    const FixedLayout L = layout_for(P, fork);
    const bool altair = fork >= FORK_ALTAIR, bellatrix = fork >= FORK_BELLATRIX, capella = fork >= FORK_CAPELLA, electra = fork >= FORK_ELECTRA;
    if (n_bytes < L.size) {
        return fail("state encoding shorter than its fixed part");
    }
    if (fork == FORK_PHASE0 && !ext_roots) return fail("phase0: the PendingAttestation list roots must be supplied");
    // variable parts, in field order
    constexpr int NV = 12;  // variable-size fields in encoding order
    const u64 off_words[NV] = {L.historical_roots_off, L.eth1_data_votes_off, L.validators_off, L.balances_off,
                               altair ? L.prev_participation_off : L.prev_attestations_off,
                               altair ? L.cur_participation_off : L.cur_attestations_off, L.inactivity_scores_off, L.payload_header_off,
                               L.historical_summaries_off, L.pending_balance_deposits_off, L.pending_partial_withdrawals_off,
                               L.pending_consolidations_off};
    u64 off[NV];
    u64 prev = L.size;
    bool first = true;
    for (int i = 0; i < NV; i++) {
        if (off_words[i] == NO_FIELD) {
            off[i] = NO_FIELD;
            continue;
        }
        off[i] = rd32(h_fixed + off_words[i]);
        if (first && off[i] != L.size) return fail("first SSZ offset does not match the fixed part");
        if (off[i] < prev || off[i] > n_bytes) return fail("SSZ offsets not monotonic");
        prev = off[i];
        first = false;
    }
    auto end_of = [&](int i) {  // where variable part i ends: the next present offset, or the end of the encoding
        for (int k = i + 1; k < NV; k++)
            if (off[k] != NO_FIELD) return off[k];
        return n_bytes;
    };
    const u64 HDR_FIXED = payload_header_fixed(fork);
    const u64 len_hroots = end_of(0) - off[0], len_votes = end_of(1) - off[1], len_vals = end_of(2) - off[2], len_bal = end_of(3) - off[3],
              len_pp = altair ? end_of(4) - off[4] : 0, len_cp = altair ? end_of(5) - off[5] : 0, len_inact = altair ? end_of(6) - off[6] : 0,
              len_hdr = bellatrix ? end_of(7) - off[7] : 0, len_hsum = capella ? end_of(8) - off[8] : 0,
              len_pbd = electra ? end_of(9) - off[9] : 0, len_ppw = electra ? end_of(10) - off[10] : 0, len_pc = electra ? end_of(11) - off[11] : 0;
    if (len_hroots % 32 || len_votes % 72 || len_vals % 121 || len_bal % 8 || len_inact % 8 || len_hsum % 64 || len_pbd % 16 || len_ppw % 24 ||
        len_pc % 16 ||
        (bellatrix && (len_hdr < HDR_FIXED || len_hdr > HDR_FIXED + 32))) {
        return fail("variable-size field has an impossible length");
    }
    if (bellatrix && h_payload_fixed && rd32(h_payload_fixed + PAYLOAD_EXTRA_DATA_OFFSET_WORD) != HDR_FIXED)
        return fail("payload header: extra_data offset does not match the fixed part");
    const u64 n_hroots = len_hroots / 32, n_votes = len_votes / 72, n_vals = len_vals / 121, n_bal = len_bal / 8,
              n_inact = len_inact / 8, n_hsum = len_hsum / 64, extra_len = bellatrix ? len_hdr - HDR_FIXED : 0;
    const u64 n_pbd = len_pbd / 16, n_ppw = len_ppw / 24, n_pc = len_pc / 16;
    if (n_hroots > P.historical_roots_limit || n_votes > P.eth1_data_votes_bound ||
        n_vals > P.validator_registry_limit || n_hsum > P.historical_roots_limit || n_pbd > P.pending_balance_deposits_limit ||
        n_ppw > P.pending_partial_withdrawals_limit || n_pc > P.pending_consolidations_limit) {
        return fail("list longer than its limit");
    }
    // chunk numbers of the field roots = field positions in the fork's container
    const u32 F_BITS = 17, F_CP0 = 18, F_INACT = 21, F_SC0 = 22, F_HDR = 24, F_NWI = 25, F_NWVI = 26, F_HSUM = 27;

    Builder B;
    std::vector<BigField>& bigs = plan.bigs;
    u32 sc_pk_root[2] = {0, 0};
    if (altair) {  // SyncCommittee container chunks: [pubkeys root, agg root]
        sc_pk_root[0] = B.alloc(2);
        sc_pk_root[1] = B.alloc(2);
    }
    auto lg = [](u64 x) { return ceil_log2_u64(x); };
    bigs.push_back({LEAF_CHUNKS, L.block_roots, 32 * P.slots_per_historical_root, P.slots_per_historical_root,
                    lg(P.slots_per_historical_root), false, 0, 5});
    bigs.push_back({LEAF_CHUNKS, L.state_roots, 32 * P.slots_per_historical_root, P.slots_per_historical_root,
                    lg(P.slots_per_historical_root), false, 0, 6});
    bigs.push_back({LEAF_CHUNKS, off[0], len_hroots, n_hroots, lg(P.historical_roots_limit), true, n_hroots, 7});
    bigs.push_back({LEAF_ETH1DATA, off[1], len_votes, n_votes, lg(P.eth1_data_votes_bound), true, n_votes, 9});
    bigs.push_back({LEAF_VALIDATORS, off[2], len_vals, n_vals, lg(P.validator_registry_limit), true, n_vals, 11});
    bigs.push_back({LEAF_CHUNKS, off[3], len_bal, (len_bal + 31) / 32, lg(P.validator_registry_limit / 4), true, n_bal, 12});
    bigs.push_back({LEAF_CHUNKS, L.randao_mixes, 32 * P.epochs_per_historical_vector, P.epochs_per_historical_vector,
                    lg(P.epochs_per_historical_vector), false, 0, 13});
    bigs.push_back({LEAF_CHUNKS, L.slashings, 8 * P.epochs_per_slashings_vector, P.epochs_per_slashings_vector / 4,
                    lg(P.epochs_per_slashings_vector / 4), false, 0, 14});
    if (altair) {
        bigs.push_back({LEAF_CHUNKS, off[4], len_pp, (len_pp + 31) / 32, lg(P.validator_registry_limit / 32), true, len_pp, 15});
        bigs.push_back({LEAF_CHUNKS, off[5], len_cp, (len_cp + 31) / 32, lg(P.validator_registry_limit / 32), true, len_cp, 16});
        bigs.push_back({LEAF_CHUNKS, off[6], len_inact, (len_inact + 31) / 32, lg(P.validator_registry_limit / 4), true, n_inact, F_INACT});
        bigs.push_back({LEAF_BYTES48, L.current_sync_committee, 48 * P.sync_committee_size, P.sync_committee_size,
                        lg(P.sync_committee_size), false, 0, sc_pk_root[0]});
        bigs.push_back({LEAF_BYTES48, L.next_sync_committee, 48 * P.sync_committee_size, P.sync_committee_size,
                        lg(P.sync_committee_size), false, 0, sc_pk_root[1]});
    } else {
        plan.ext_chunks.push_back({15u, 0u});  // previous_epoch_attestations, current_epoch_attestations: roots from the caller
        plan.ext_chunks.push_back({16u, 32u});
    }
    if (capella) bigs.push_back({LEAF_PAIR64, off[8], len_hsum, n_hsum, lg(P.historical_roots_limit), true, n_hsum, F_HSUM});
    if (electra) {  // fields 34, 35, 36: lists of two- / three-uint64 containers
        bigs.push_back({LEAF_U64X2, off[9], len_pbd, n_pbd, lg(P.pending_balance_deposits_limit), true, n_pbd, 34});
        bigs.push_back({LEAF_U64X3, off[10], len_ppw, n_ppw, lg(P.pending_partial_withdrawals_limit), true, n_ppw, 35});
        bigs.push_back({LEAF_U64X2, off[11], len_pc, n_pc, lg(P.pending_consolidations_limit), true, n_pc, 36});
    }

    // ---- small fields: gathers + jobs ----------------------------------------------------------
    // basic fields: the root is the zero-padded chunk itself
    B.gather(L.genesis_time, 8, 0);
    B.gather(L.genesis_validators_root, 32, 1);
    B.gather(L.slot, 8, 2);
    B.gather(L.eth1_deposit_index, 8, 10);
    B.gather(L.justification_bits, 1, F_BITS);
    if (capella) {
        B.gather(L.next_withdrawal_index, 8, F_NWI);
        B.gather(L.next_withdrawal_validator_index, 8, F_NWVI);
    }
    if (electra)  // fields 28 .. 33: six uint64
        for (u32 k = 0; k < 6; k++) B.gather(L.deposit_receipts_start_index + 8 * k, 8, 28 + k);
    {   // Fork: previous_version[4], current_version[4], epoch u64
        u32 c0 = B.alloc(3);
        B.gather(L.fork, 4, c0);
        B.gather(L.fork + 4, 4, c0 + 1);
        B.gather(L.fork + 8, 8, c0 + 2);
        B.job(0, c0, 3, 2, 3);
    }
    {   // BeaconBlockHeader: slot, proposer_index, parent_root, state_root, body_root
        u32 c0 = B.alloc(5);
        B.gather(L.latest_block_header, 8, c0);
        B.gather(L.latest_block_header + 8, 8, c0 + 1);
        for (u32 k = 0; k < 3; k++) B.gather(L.latest_block_header + 16 + 32 * k, 32, c0 + 2 + k);
        B.job(0, c0, 5, 3, 4);
    }
    {   // Eth1Data: deposit_root, deposit_count, block_hash
        u32 c0 = B.alloc(3);
        B.gather(L.eth1_data, 32, c0);
        B.gather(L.eth1_data + 32, 8, c0 + 1);
        B.gather(L.eth1_data + 40, 32, c0 + 2);
        B.job(0, c0, 3, 2, 8);
    }
    const u64 cps[3] = {L.prev_justified, L.cur_justified, L.finalized};
    for (u32 k = 0; k < 3; k++) {  // Checkpoint: epoch, root
        u32 c0 = B.alloc(2);
        B.gather(cps[k], 8, c0);
        B.gather(cps[k] + 8, 32, c0 + 1);
        B.job(0, c0, 2, 1, F_CP0 + k);
    }
    const u64 scs[2] = {L.current_sync_committee, L.next_sync_committee};
    for (u32 k = 0; k < 2 && altair; k++) {  // SyncCommittee: htr(Vector<PublicKey>) (big pass), htr(aggregate_public_key)
        u32 c0 = B.alloc(2);
        u64 agg = scs[k] + 48 * P.sync_committee_size;
        B.gather(agg, 32, c0);
        B.gather(agg + 32, 16, c0 + 1);
        B.job(0, c0, 2, 1, sc_pk_root[k] + 1);
        B.job(1, sc_pk_root[k], 2, 1, F_SC0 + k);
    }
    if (bellatrix) {   // ExecutionPayloadHeader: 14 (bellatrix), 15 (capella) or 17 (deneb) fields
        const u64 h = off[7];
        const u32 nf = fork == FORK_BELLATRIX ? 14u : fork == FORK_CAPELLA ? 15u : fork == FORK_ELECTRA ? 19u : 17u;
        u32 f = B.alloc(nf);
        B.gather(h + 0, 32, f + 0);      // parent_hash
        B.gather(h + 32, 20, f + 1);     // fee_recipient
        B.gather(h + 52, 32, f + 2);     // state_root
        B.gather(h + 84, 32, f + 3);     // receipts_root
        u32 bloom = B.alloc(8);          // logs_bloom: 256 bytes = 8 chunks
        for (u32 k = 0; k < 8; k++) B.gather(h + 116 + 32 * k, 32, bloom + k);
        B.job(0, bloom, 8, 3, f + 4);
        B.gather(h + 372, 32, f + 5);    // prev_randao
        B.gather(h + 404, 8, f + 6);     // block_number
        B.gather(h + 412, 8, f + 7);     // gas_limit
        B.gather(h + 420, 8, f + 8);     // gas_used
        B.gather(h + 428, 8, f + 9);     // timestamp
        u32 extra = B.alloc(1);          // extra_data: ByteList<32> -> one chunk, mix in length
        B.gather(h + HDR_FIXED, (u32)extra_len, extra);
        B.job(0, extra, extra_len ? 1 : 0, 0, f + 10, true, extra_len);
        B.gather(h + 440, 32, f + 11);   // base_fee_per_gas (U256 LE)
        B.gather(h + 472, 32, f + 12);   // block_hash
        B.gather(h + 504, 32, f + 13);   // transactions_root
        if (capella) B.gather(h + 536, 32, f + 14);  // withdrawals_root
        if (fork >= FORK_DENEB) {
            B.gather(h + 568, 8, f + 15);  // blob_gas_used
            B.gather(h + 576, 8, f + 16);  // excess_blob_gas
        }
        if (electra) {
            B.gather(h + 584, 32, f + 17);  // deposit_receipts_root
            B.gather(h + 616, 32, f + 18);  // withdrawal_requests_root
        }
        B.job(1, f, nf, lg(nf), F_HDR);
    }
    const u32 root_chunk = B.alloc(1);
    B.job(2, 0, state_field_count(fork), ceil_log2_u64(state_field_chunks(fork)), root_chunk);

    plan.gathers = B.gathers;
    for (int l = 0; l < 3; l++) plan.jobs[l] = B.jobs[l];
    plan.n_small_chunks = B.next_chunk;
    plan.root_chunk = root_chunk;
    plan.small_hashes = B.hashes;
    plan.payload_header_off = bellatrix ? off[7] : NO_FIELD;
    return true;
}
inline bool build_state_plan_deneb(const u8* h_fixed, u64 n_bytes, int preset, StatePlan& plan) {
    return build_state_plan(FORK_DENEB, h_fixed, n_bytes, preset, plan);
}

}  // namespace ecg
