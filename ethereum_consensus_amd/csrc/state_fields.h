// Field-addressed writes to a resident BeaconState (round 6): pure host arithmetic, no hashing and no HIP calls -- the product
// (state_deneb.hip) runs it over the device-resident encoding, tests/hostsim runs the very same code over a host byte array.
//
// The reference's state transition names what it changes by FIELD and INDEX, never by byte offset in a serialization:
//   state.balances[index] += delta                         phase0/helpers.rs:979-1030 (increase_balance / decrease_balance)
//   state.validators.push(..); state.balances.push(..)     phase0/block_processing.rs:317-349 (add_validator_to_registry;
//                                                           altair/block_processing.rs:192-213 also pushes two participation
//                                                           flags and an inactivity score)
//   state.current_epoch_participation[index] = flags        altair/block_processing.rs:98-170 (process_attestation)
//   state.eth1_data_votes.push(vote) / .clear()             phase0/block_processing.rs:689-700, phase0/epoch_processing.rs
//   state.state_roots[slot % N] = root; state.slot += 1     phase0/slot_processing.rs:58-86
// Up to round 5 the C ABI took absolute byte offsets (ecgpu_resident_state_patch), so the (field, index) -> offset arithmetic --
// which changes whenever a list in front of the field grows -- lived in the never-compiled Rust shim.  It lives here now:
// a field is named by its POSITION in the fork's BeaconState container (phase0/beacon_state.rs:50-88, altair/beacon_state.rs:
// 13-55, ... electra/beacon_state.rs:73-145: positions are stable from altair on; phase0 holds its two PendingAttestation lists
// at 15 and 16), a write by a byte offset INSIDE that field, and everything is resolved against the encoding as it is when
// the bytes are applied -- after the length changes queued before it.
//
// Semantics: the calls behave as if each were applied in program order.  Writes and pushes are QUEUED on the host (a slot's
// 4 096 balance writes must not be 4 096 launches) and travel in one block at the next root / flush; rare operations
// (truncate, replacing a whole variable-size field, the participation rotation) flush the queue and run at once.
#pragma once
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "state_plan.h"

namespace ecg {

constexpr u32 FIELD_NOT_VARIABLE = 0xffffffffu;
constexpr int N_STATE_VAR_FIELDS = 12;  // variable-size fields in encoding order (build_state_plan's table)

// one field of the fork's container, located in the CURRENT encoding
struct FieldLoc {
    bool variable = false;
    u32 var_index = FIELD_NOT_VARIABLE;  // position among the variable-size fields (the ECGPU_STATE_* numbering of include/ecgpu.h)
    u64 start = 0, len = 0;              // byte range in the encoding
    u32 elem = 0;                        // element size of a list / vector; the field's size for containers and basic fields; 0: variable-size elements
    u64 limit_bytes = 0;                 // lists: elem * limit; 0 otherwise
};

// static description of position `field`: fixed offset + size, or the variable-field index
struct FieldStatic {
    bool present = false, variable = false;
    u64 off = 0, size = 0;  // fixed-size fields: byte range in the fixed part
    u32 var_index = FIELD_NOT_VARIABLE, elem = 0;
    u64 limit = 0;          // element limit of a list
};

inline FieldStatic field_static(int fork, int preset, u32 field) {
    FieldStatic f;
    if (preset < 0 || preset > 1 || fork < FORK_PHASE0 || fork > FORK_LAST || field >= state_field_count(fork)) return f;
    const Preset& P = STATE_PRESETS[preset];
    const FixedLayout L = layout_for(P, fork);
    auto fixed = [&](u64 off, u64 size, u32 elem) {
        f.present = true, f.off = off, f.size = size, f.elem = elem;
    };
    auto var = [&](u32 vi, u32 elem, u64 limit) {
        f.present = true, f.variable = true, f.var_index = vi, f.elem = elem, f.limit = limit;
    };
    const u64 sc = 48 * P.sync_committee_size + 48;
    switch (field) {
        case 0: fixed(L.genesis_time, 8, 8); break;
        case 1: fixed(L.genesis_validators_root, 32, 32); break;
        case 2: fixed(L.slot, 8, 8); break;
        case 3: fixed(L.fork, 16, 16); break;
        case 4: fixed(L.latest_block_header, 112, 112); break;
        case 5: fixed(L.block_roots, 32 * P.slots_per_historical_root, 32); break;
        case 6: fixed(L.state_roots, 32 * P.slots_per_historical_root, 32); break;
        case 7: var(0, 32, P.historical_roots_limit); break;
        case 8: fixed(L.eth1_data, 72, 72); break;
        case 9: var(1, 72, P.eth1_data_votes_bound); break;
        case 10: fixed(L.eth1_deposit_index, 8, 8); break;
        case 11: var(2, 121, P.validator_registry_limit); break;
        case 12: var(3, 8, P.validator_registry_limit); break;
        case 13: fixed(L.randao_mixes, 32 * P.epochs_per_historical_vector, 32); break;
        case 14: fixed(L.slashings, 8 * P.epochs_per_slashings_vector, 8); break;
        case 15: var(4, fork == FORK_PHASE0 ? 0u : 1u, fork == FORK_PHASE0 ? 0 : P.validator_registry_limit); break;
        case 16: var(5, fork == FORK_PHASE0 ? 0u : 1u, fork == FORK_PHASE0 ? 0 : P.validator_registry_limit); break;
        case 17: fixed(L.justification_bits, 1, 1); break;
        case 18: fixed(L.prev_justified, 40, 40); break;
        case 19: fixed(L.cur_justified, 40, 40); break;
        case 20: fixed(L.finalized, 40, 40); break;
        case 21: var(6, 8, P.validator_registry_limit); break;
        case 22: fixed(L.current_sync_committee, sc, 48); break;
        case 23: fixed(L.next_sync_committee, sc, 48); break;
        case 24: var(7, 0, 0); break;  // latest_execution_payload_header: one variable-size container (extra_data)
        case 25: fixed(L.next_withdrawal_index, 8, 8); break;
        case 26: fixed(L.next_withdrawal_validator_index, 8, 8); break;
        case 27: var(8, 64, P.historical_roots_limit); break;
        case 28: case 29: case 30: case 31: case 32: case 33:
            fixed(L.deposit_receipts_start_index + 8 * (field - 28), 8, 8); break;
        case 34: var(9, 16, P.pending_balance_deposits_limit); break;
        case 35: var(10, 24, P.pending_partial_withdrawals_limit); break;
        case 36: var(11, 16, P.pending_consolidations_limit); break;
        default: break;
    }
    return f;
}

// positions of the offset words of the variable-size fields in the fixed part (NO_FIELD: not in this fork), encoding order
inline void state_offset_words(int fork, int preset, u64 out[N_STATE_VAR_FIELDS]) {
    const FixedLayout L = layout_for(STATE_PRESETS[preset], fork);
    const bool altair = fork >= FORK_ALTAIR;
    const u64 w[N_STATE_VAR_FIELDS] = {L.historical_roots_off, L.eth1_data_votes_off, L.validators_off, L.balances_off,
                                       altair ? L.prev_participation_off : L.prev_attestations_off,
                                       altair ? L.cur_participation_off : L.cur_attestations_off, L.inactivity_scores_off, L.payload_header_off,
                                       L.historical_summaries_off, L.pending_balance_deposits_off, L.pending_partial_withdrawals_off,
                                       L.pending_consolidations_off};
    for (int i = 0; i < N_STATE_VAR_FIELDS; i++) out[i] = w[i];
}

// the static part of the above for every (fork, preset, position), built once: a hook of the state transition is ONE call per
// element written (rust/ecgpu-shim StateMirror) -- 8 192 of them per slot -- and must cost a table lookup, not a layout walk
struct FieldTables {
    FieldStatic f[FORK_LAST + 1][2][STATE_MAX_FIELD_CHUNKS];
    u64 words[FORK_LAST + 1][2][N_STATE_VAR_FIELDS];
    FieldTables() {
        for (int fork = 0; fork <= FORK_LAST; fork++)
            for (int preset = 0; preset < 2; preset++) {
                for (u32 k = 0; k < STATE_MAX_FIELD_CHUNKS; k++) f[fork][preset][k] = field_static(fork, preset, k);
                state_offset_words(fork, preset, words[fork][preset]);
            }
    }
};
inline const FieldTables& field_tables() {
    static const FieldTables T;
    return T;
}

// where field `field` lies in the encoding described by `h_fixed` (its fixed part) and `n_bytes` (its length) NOW
inline bool locate_field(int fork, int preset, const u8* h_fixed, u64 n_bytes, u32 field, FieldLoc& out) {
    if (preset < 0 || preset > 1 || fork < FORK_PHASE0 || fork > FORK_LAST || field >= STATE_MAX_FIELD_CHUNKS) return false;
    const FieldTables& T = field_tables();
    const FieldStatic& f = T.f[fork][preset][field];
    if (!f.present) return false;
    out = FieldLoc{};
    out.elem = f.elem;
    if (!f.variable) {
        out.start = f.off, out.len = f.size;
        return true;
    }
    const u64* words = T.words[fork][preset];
    if (words[f.var_index] == NO_FIELD) return false;
    out.variable = true;
    out.var_index = f.var_index;
    out.start = rd32(h_fixed + words[f.var_index]);
    u64 end = n_bytes;
    for (int k = (int)f.var_index + 1; k < N_STATE_VAR_FIELDS; k++)
        if (words[k] != NO_FIELD) {
            end = rd32(h_fixed + words[k]);
            break;
        }
    if (end < out.start || end > n_bytes) return false;
    out.len = end - out.start;
    out.limit_bytes = (u64)f.elem * f.limit;
    return true;
}

enum FieldResize { FIELD_APPEND = 0, FIELD_TRUNCATE = 1, FIELD_REPLACE = 2 };

// The queue.  `Sink` is what the bytes are applied to:
//   int fork() / preset();  const u8* fixed();  u64 size();                       the encoding as it is now
//   int apply_patches(const u64* offsets, const u64* data_off, const u8* data, u32 n)   non-overlapping overwrites, absolute offsets
//   int apply_resize(u32 var_index, const u8* data, u64 add_len, u64 keep_len, FieldResize mode)
//   int apply_rotate(u64 prev_start, u64 cur_start, u64 len)                       previous <- current, current <- 0
//   void fail(const char* msg)
// Every method returns 0 or the (negative) code the C entry hands back.
template <class Sink>
struct FieldWriter {
    struct Write {
        u32 field;
        u64 off, len, src;  // bytes blob[src .. src + len) -> field bytes [off, off + len)
    };
    std::vector<Write> writes;
    std::vector<u8> blob;
    std::vector<u8> pushed[N_STATE_VAR_FIELDS];  // elements appended to list var_index since the last flush
    // Which ELEMENTS of a field the queued writes touch: a bit per element, set when a write is queued.  A slot's 8 192 writes hit
    // 8 192 different balances and flags: as long as no element of a field is touched twice its writes are disjoint as they
    // stand and go to the sink as queued (30 ns each); only a field in which an element WAS touched twice (`collided`: two members
    // of one validator record, the same balance written twice) is resolved by interval subtraction (flush_writes).  Conservative:
    // every write marks every element it overlaps.
    struct Marks {
        std::vector<u64> bits;
        std::vector<u32> touched;  // words of `bits` that are non-zero (to clear them without sweeping the bitmap)
        bool collided = false;
        void mark(u64 e0, u64 e1) {
            if (collided) return;
            if (e1 - e0 > 512) {  // (a long range: not worth a bit per element)
                collided = true;
                return;
            }
            if ((e1 >> 6) >= bits.size()) bits.resize((e1 >> 6) + 1 + (bits.size() >> 1), 0);
            for (u64 e = e0; e <= e1; e++) {
                u64& w = bits[e >> 6];
                const u64 b = 1ull << (e & 63);
                if (w & b) {
                    collided = true;
                    return;
                }
                if (!w) touched.push_back((u32)(e >> 6));
                w |= b;
            }
        }
        void clear() {
            for (u32 w : touched) bits[w] = 0;
            touched.clear();
            collided = false;
        }
    };
    Marks marks[STATE_MAX_FIELD_CHUNKS];
    static constexpr int BAD = -3;                // ECGPU_ERR_BAD_ARG
    static constexpr u64 DIRECT_BYTES = 1u << 16;  // a write this large is not copied into the queue: flush, then apply from where it lies

    bool pending() const {
        if (!writes.empty()) return true;
        for (const auto& p : pushed)
            if (!p.empty()) return true;
        return false;
    }
    u64 pushed_bytes() const {
        u64 t = 0;
        for (const auto& p : pushed) t += p.size();
        return t;
    }
    // the field as the caller sees it: applied length + what is queued behind it
    bool locate(Sink& s, u32 field, FieldLoc& loc, u64& seen_len) const {
        if (!locate_field(s.fork(), s.preset(), s.fixed(), s.size(), field, loc)) return false;
        seen_len = loc.len + (loc.variable ? pushed[loc.var_index].size() : 0);
        return true;
    }

    int write(Sink& s, u32 field, u64 off, const u8* data, u64 n) {
        FieldLoc loc;
        u64 seen;
        if (!locate(s, field, loc, seen)) return s.fail("no such field in this fork"), BAD;
        if (loc.variable && loc.elem == 0 && loc.var_index != 7) return s.fail("a list of variable-size elements changes by replacement only"), BAD;
        if (off > seen || n > seen - off) return s.fail("write outside the field"), BAD;
        if (!n) return 0;
        if (!data) return BAD;
        if (off + n > loc.len) {  // (part of) it lands in elements that are still queued: edit them where they are
            const u64 lo = off > loc.len ? off : loc.len;
            std::memcpy(pushed[loc.var_index].data() + (lo - loc.len), data + (lo - off), off + n - lo);
            n = lo - off;
            if (!n) return 0;
        }
        if (n >= DIRECT_BYTES) {
            int rc = flush_writes(s);
            if (rc) return rc;
            const u64 o = loc.start + off, d[2] = {0, n};
            return s.apply_patches(&o, d, data, 1);
        }
        const u64 gran = loc.elem ? loc.elem : 32;  // (any granularity is correct; the element's is the one without false collisions)
        marks[field].mark(off / gran, (off + n - 1) / gran);
        writes.push_back({field, off, n, (u64)blob.size()});
        blob.insert(blob.end(), data, data + n);
        return 0;
    }
    int push(Sink& s, u32 field, const u8* data, u64 n) {
        FieldLoc loc;
        u64 seen;
        if (!locate(s, field, loc, seen) || !loc.variable || !loc.elem) return s.fail("not a list of fixed-size elements of this fork"), BAD;
        if (n % loc.elem || (n && !data)) return s.fail("not a whole number of elements"), BAD;
        if (seen + n > loc.limit_bytes) return s.fail("list longer than its limit"), BAD;
        pushed[loc.var_index].insert(pushed[loc.var_index].end(), data, data + n);
        return 0;
    }
    int truncate(Sink& s, u32 field, u64 keep) {
        FieldLoc loc;
        u64 seen;
        if (!locate(s, field, loc, seen) || !loc.variable || !loc.elem) return s.fail("not a list of fixed-size elements of this fork"), BAD;
        if (keep > seen || keep % loc.elem) return s.fail("truncate: not a shorter whole number of elements"), BAD;
        if (keep >= loc.len) {  // only queued elements go
            pushed[loc.var_index].resize(keep - loc.len);
            return 0;
        }
        pushed[loc.var_index].clear();
        int rc = flush_writes(s);  // (writes behind `keep` die with their bytes: applied first, cut off next)
        if (rc) return rc;
        return s.apply_resize(loc.var_index, nullptr, 0, keep, FIELD_TRUNCATE);
    }
    // the whole field: fixed-size fields (and a same-length list) are a write; a list of another length, the payload header
    // with another extra_data and phase0's attestation lists are exchanged on the spot
    int set(Sink& s, u32 field, const u8* data, u64 n) {
        FieldLoc loc;
        u64 seen;
        if (!locate(s, field, loc, seen)) return s.fail("no such field in this fork"), BAD;
        if (n && !data) return BAD;
        if (!loc.variable) {
            if (n != loc.len) return s.fail("a fixed-size field keeps its size"), BAD;
            return write(s, field, 0, data, n);
        }
        if (loc.elem && n % loc.elem) return s.fail("not a whole number of elements"), BAD;
        if (loc.elem && n > loc.limit_bytes) return s.fail("list longer than its limit"), BAD;
        if (loc.elem && n == seen && loc.var_index != 7) return write(s, field, 0, data, n);
        if (loc.var_index == 7) {  // the payload header: what the reference's deserializer accepts (fixed part, <= 32 bytes of extra_data, its offset word)
            const u64 hf = payload_header_fixed(s.fork());
            if (n < hf || n > hf + 32 || rd32(data + PAYLOAD_EXTRA_DATA_OFFSET_WORD) != hf) return s.fail("payload header: bad length or extra_data offset"), BAD;
        }
        // writes to this field queued so far are overwritten by the replacement: drop them, apply the others first
        std::vector<Write> keep_w;
        for (const Write& w : writes)
            if (w.field != field) keep_w.push_back(w);
        writes.swap(keep_w);
        marks[field].clear();
        pushed[loc.var_index].clear();
        int rc = flush_writes(s);
        if (rc) return rc;
        return s.apply_resize(loc.var_index, data, n, 0, FIELD_REPLACE);
    }
    // process_participation_flag_updates (altair/epoch_processing.rs): previous_epoch_participation = current; current = zeros
    int rotate_participation(Sink& s) {
        if (s.fork() < FORK_ALTAIR) return s.fail("phase0 rotates its attestation lists by replacement"), BAD;
        int rc = flush(s);
        if (rc) return rc;
        FieldLoc p, c;
        if (!locate_field(s.fork(), s.preset(), s.fixed(), s.size(), 15, p) || !locate_field(s.fork(), s.preset(), s.fixed(), s.size(), 16, c) || p.len != c.len)
            return s.fail("participation lists of different lengths"), BAD;
        return p.len ? s.apply_rotate(p.start, c.start, p.len) : 0;
    }
    // add_validator_to_registry: one element on each registry-sized list
    int add_validator(Sink& s, const u8* record121, u64 balance) {
        u8 bal[8], zero[8] = {};
        for (int i = 0; i < 8; i++) bal[i] = (u8)(balance >> (8 * i));
        // refuse before anything is queued (the lists stay the same length as each other)
        FieldLoc loc;
        u64 seen;
        if (!locate(s, 11, loc, seen)) return BAD;
        if (seen + 121 > loc.limit_bytes) return s.fail("list longer than its limit"), BAD;
        int rc = push(s, 11, record121, 121);
        if (!rc) rc = push(s, 12, bal, 8);
        if (!rc && s.fork() >= FORK_ALTAIR) {
            rc = push(s, 15, zero, 1);
            if (!rc) rc = push(s, 16, zero, 1);
            if (!rc) rc = push(s, 21, zero, 8);
        }
        return rc;
    }

    // queued pushes, then queued writes
    int flush(Sink& s) {
        for (u32 vi = 0; vi < (u32)N_STATE_VAR_FIELDS; vi++)
            if (!pushed[vi].empty()) {
                std::vector<u8> data;
                data.swap(pushed[vi]);
                int rc = s.apply_resize(vi, data.data(), data.size(), 0, FIELD_APPEND);
                if (rc) return drop(), rc;
            }
        return flush_writes(s);
    }
    void drop() {
        writes.clear();
        blob.clear();
        for (auto& p : pushed) p.clear();
        for (auto& m : marks) m.clear();
    }

    // the queued writes as ONE set of non-overlapping patches: where two writes cover the same byte the later one wins
    int flush_writes(Sink& s) {
        if (writes.empty()) return 0;
        std::vector<u64> offsets, data_off;
        std::vector<u8> data;
        offsets.reserve(writes.size());
        data_off.reserve(writes.size() + 1);
        data.reserve(blob.size());
        data_off.push_back(0);
        FieldLoc locs[STATE_MAX_FIELD_CHUNKS];
        bool located[STATE_MAX_FIELD_CHUNKS] = {};
        auto loc_of = [&](u32 field, const FieldLoc*& out) {
            if (!located[field]) {
                if (!locate_field(s.fork(), s.preset(), s.fixed(), s.size(), field, locs[field])) return false;
                located[field] = true;
            }
            out = &locs[field];
            return true;
        };
        bool any_collided = false;
        // fields whose elements were each touched at most once: the writes are disjoint as queued
        for (const Write& w : writes) {
            if (marks[w.field].collided) {
                any_collided = true;
                continue;
            }
            const FieldLoc* loc;
            if (!loc_of(w.field, loc)) return drop(), s.fail("field vanished"), BAD;
            if (w.off + w.len > loc->len) return drop(), s.fail("queued write outside the field"), BAD;  // (cannot happen: checked when queued)
            offsets.push_back(loc->start + w.off);
            data.insert(data.end(), blob.begin() + w.src, blob.begin() + (w.src + w.len));
            data_off.push_back(data.size());
        }
        // the others: newest first, each write gives the sink only the bytes no later write of the same field has claimed
        if (any_collided) {
            std::map<u32, std::map<u64, u64>> covered;  // field -> disjoint [start, end) ranges already claimed by later writes
            for (size_t k = writes.size(); k-- > 0;) {
                const Write& w = writes[k];
                if (!marks[w.field].collided) continue;
                const FieldLoc* locp;
                if (!loc_of(w.field, locp)) return drop(), s.fail("field vanished"), BAD;
                const FieldLoc& loc = *locp;
                if (w.off + w.len > loc.len) return drop(), s.fail("queued write outside the field"), BAD;
                std::map<u64, u64>& cov = covered[w.field];
                // pieces of [w.off, w.off + w.len) not in `cov`
                u64 pos = w.off;
                const u64 end = w.off + w.len;
                auto c = cov.upper_bound(pos);
                if (c != cov.begin()) {
                    auto p = std::prev(c);
                    if (p->second > pos) pos = p->second < end ? p->second : end;
                }
                while (pos < end) {
                    const u64 stop = (c != cov.end() && c->first < end) ? c->first : end;
                    if (stop > pos) {
                        offsets.push_back(loc.start + pos);
                        data.insert(data.end(), blob.begin() + (w.src + (pos - w.off)), blob.begin() + (w.src + (stop - w.off)));
                        data_off.push_back(data.size());
                    }
                    if (c == cov.end() || c->first >= end) break;
                    pos = c->second < end ? c->second : end;
                    ++c;
                }
                // claim [w.off, end): merge with what it touches
                u64 lo = w.off, hi = end;
                auto a = cov.lower_bound(lo);
                if (a != cov.begin() && std::prev(a)->second >= lo) --a;
                while (a != cov.end() && a->first <= hi) {
                    if (a->first < lo) lo = a->first;
                    if (a->second > hi) hi = a->second;
                    a = cov.erase(a);
                }
                cov[lo] = hi;
            }
        }
        writes.clear();
        blob.clear();
        for (auto& m : marks) m.clear();
        if (offsets.empty()) return 0;
        return s.apply_patches(offsets.data(), data_off.data(), data.data(), (u32)offsets.size());
    }
};

}  // namespace ecg
