// Merkle proofs and generalized indices over the generic SSZ type description (include/ecgpu.h ecgpu_ssz_type): what
// ssz_rs's `Prove` / `GeneralizedIndexable` give every derived container of the reference -- exercised at
// /root/reference/spec-tests/runners/light_client.rs:42-69 (prove + verify of sync-committee / finality / execution
// branches), ethereum-consensus/src/deneb/blob_sidecar.rs:47-64 (generalized_index of blob_kzg_commitments[i]) and
// deneb/beacon_block.rs:139-154 (the indices 27, 221184..221189).  SURVEY.md 8f rank 4, second half.
//
// A proof is latency work (tens of hash64 along one path), so this file is orchestration: the walk down the path is host
// offset arithmetic; every hash64 runs on the GPU through the existing engines -- ecgpu_htr_ssz for the roots of the
// siblings' subtrees that are SSZ objects, merkleize_device for sibling subtrees inside one object's chunk tree.
#include <cstring>
#include <vector>

#include "merkle_driver.h"
#include "ssz_plan.h"

namespace ecg {
int htr_ssz_impl(const ecgpu_ssz_type* types, uint32_t n_types, const uint32_t* fields, uint32_t n_field_refs, uint32_t root_type,
                 const uint8_t* ssz, uint64_t n_bytes, uint8_t root[32], bool allow_internal);  // ssz_generic.hip


namespace {

constexpr u64 kVariable = ~0ull - 1;

struct Walker {
    const ecgpu_ssz_type* T;
    u32 nT;
    const u32* F;
    u32 nF;
    std::string error;

    int fail(const char* m) {
        if (error.empty()) error = m;
        return ECGPU_ERR_BAD_ARG;
    }
    u64 fixed_size(u32 ti) const {
        const ecgpu_ssz_type& t = T[ti];
        switch (t.kind) {
            case ECGPU_SSZ_UINT:
            case ECGPU_SSZ_BYTEVECTOR: return t.param;
            case ECGPU_SSZ_BITVECTOR: return (t.param + 7) / 8;
            case ECGPU_SSZ_VECTOR: {
                const u64 e = fixed_size(t.elem);
                return e == kVariable ? kVariable : e * t.param;
            }
            case ECGPU_SSZ_CONTAINER: {
                u64 s = 0;
                for (u32 k = 0; k < t.n_fields; k++) {
                    const u64 e = fixed_size(F[t.first_field + k]);
                    if (e == kVariable) return kVariable;
                    s += e;
                }
                return s;
            }
            default: return kVariable;
        }
    }
    bool valid_types() {
        for (u32 i = 0; i < nT; i++) {
            const ecgpu_ssz_type& t = T[i];
            if (t.kind > ECGPU_SSZ_CONTAINER) return false;
            if ((t.kind == ECGPU_SSZ_VECTOR || t.kind == ECGPU_SSZ_LIST) && t.elem >= i) return false;
            if (t.kind == ECGPU_SSZ_CONTAINER) {
                if ((u64)t.first_field + t.n_fields > nF || t.n_fields == 0) return false;
                for (u32 k = 0; k < t.n_fields; k++)
                    if (F[t.first_field + k] >= i) return false;
            }
        }
        return true;
    }
    // chunk count (= leaf limit) of the data tree of type ti, and whether its root mixes a length in
    void tree_shape(u32 ti, u64& limit_chunks, bool& mix) const {
        const ecgpu_ssz_type& t = T[ti];
        mix = t.kind == ECGPU_SSZ_LIST || t.kind == ECGPU_SSZ_BYTELIST || t.kind == ECGPU_SSZ_BITLIST;
        switch (t.kind) {
            case ECGPU_SSZ_UINT: limit_chunks = 1; break;
            case ECGPU_SSZ_BYTEVECTOR:
            case ECGPU_SSZ_BYTELIST: limit_chunks = (t.param + 31) / 32; break;
            case ECGPU_SSZ_BITVECTOR:
            case ECGPU_SSZ_BITLIST: limit_chunks = (t.param + 255) / 256; break;
            case ECGPU_SSZ_VECTOR:
            case ECGPU_SSZ_LIST:
                limit_chunks = T[t.elem].kind == ECGPU_SSZ_UINT ? (t.param * T[t.elem].param + 31) / 32 : t.param;
                break;
            default: limit_chunks = t.n_fields;
        }
        if (limit_chunks == 0) limit_chunks = 1;
    }
    // one step of a path inside type ti: the chunk position of the child in ti's data tree, the child's type (or ~0u when
    // the step lands on a packed chunk / the length node), following ssz_rs's generalized_index rules
    int step(u32 ti, u64 elem, u64& g, u32& child_type, bool& is_len) {
        const ecgpu_ssz_type& t = T[ti];
        u64 limit;
        bool mix;
        tree_shape(ti, limit, mix);
        const u32 depth = ceil_log2_u64(limit);
        is_len = false;
        child_type = ~0u;
        if (elem == ECGPU_SSZ_PATH_LENGTH) {
            if (!mix) return fail("only lists have a length node");
            g = g * 2 + 1;
            is_len = true;
            return 0;
        }
        if (mix) g = g * 2;
        u64 pos;
        switch (t.kind) {
            case ECGPU_SSZ_CONTAINER:
                if (elem >= t.n_fields) return fail("field position outside the container");
                pos = elem;
                child_type = F[t.first_field + elem];
                break;
            case ECGPU_SSZ_VECTOR:
            case ECGPU_SSZ_LIST:
                if (elem >= t.param) return fail("element index outside the type");
                if (T[t.elem].kind == ECGPU_SSZ_UINT) {
                    pos = elem * T[t.elem].param / 32;
                } else {
                    pos = elem;
                    child_type = t.elem;
                }
                break;
            case ECGPU_SSZ_BYTEVECTOR:
            case ECGPU_SSZ_BYTELIST:
                if (elem >= t.param) return fail("byte index outside the type");
                pos = elem / 32;
                break;
            case ECGPU_SSZ_BITVECTOR:
            case ECGPU_SSZ_BITLIST:
                if (elem >= t.param) return fail("bit index outside the type");
                pos = elem / 256;
                break;
            default: return fail("a basic value has no children");
        }
        if (depth >= 63 || g >> (62 - depth)) return fail("generalized index does not fit 64 bits");
        g = (g << depth) + pos;
        return 0;
    }
};

static u32 rd32p(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }

// branch of chunk `index` in merkleize(chunks[0..n), limit) on stream s: sibling subtree roots, bottom-up, to d_branch
static int proof_chunks_device(hipStream_t s, ThreadCtx* c, const u8* d_chunks, u64 n, u64 limit, u64 index, u8* d_branch, u8* ws) {
    const u32 depth = ceil_log2_u64(limit);
    u64 hc = 0;
    for (u32 l = 0; l < depth; l++) {
        const u64 sib = (index >> l) ^ 1, start = sib << l;
        const u64 cnt = start >= n ? 0 : ((n - start) < (1ull << l) ? n - start : (1ull << l));
        int rc = merkleize_device(s, LEAF_CHUNKS, d_chunks + 32 * (start < n ? start : 0), 32 * cnt, cnt, l, false, 0, d_branch + 32ull * l, ws, &hc);
        if (rc) return rc;
    }
    c->last_hash64 += hc;
    return ECGPU_SUCCESS;
}

}  // namespace

}  // namespace ecg

using namespace ecg;

extern "C" {

int ecgpu_ssz_generalized_index(const ecgpu_ssz_type* types, uint32_t n_types, const uint32_t* fields, uint32_t n_field_refs,
                                uint32_t root_type, const uint64_t* path, uint32_t path_len, uint64_t* gindex) {
    if (!types || root_type >= n_types || (path_len && !path) || !gindex || (n_field_refs && !fields)) return ECGPU_ERR_BAD_ARG;
    Walker w{types, n_types, fields, n_field_refs, {}};
    if (!w.valid_types()) {
        set_last_error("bad type description");
        return ECGPU_ERR_BAD_ARG;
    }
    u64 g = 1;
    u32 ti = root_type;
    for (u32 k = 0; k < path_len; k++) {
        if (ti == ~0u) {
            set_last_error("path continues below a leaf");
            return ECGPU_ERR_BAD_ARG;
        }
        u32 child;
        bool is_len;
        if (w.step(ti, path[k], g, child, is_len)) {
            set_last_error(w.error);
            return ECGPU_ERR_BAD_ARG;
        }
        ti = child;
    }
    *gindex = g;
    return ECGPU_SUCCESS;
}

int ecgpu_merkle_proof(const uint8_t* chunks, uint64_t n_chunks, uint64_t limit_chunks, uint64_t index, uint8_t* branch) {
    int rc = ensure_init();
    if (rc) return rc;
    const u64 limit = limit_chunks ? limit_chunks : (n_chunks ? n_chunks : 1);
    if ((!chunks && n_chunks) || !branch || n_chunks > limit || index >= (1ull << ceil_log2_u64(limit)) || ceil_log2_u64(limit) > 62)
        return ECGPU_ERR_BAD_ARG;
    const u32 depth = ceil_log2_u64(limit);
    if (depth == 0) return ECGPU_SUCCESS;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    Arena& ar = c->arena(s);
    ar.reset();
    rc = ar.reserve(32 * n_chunks + 32ull * depth + merkle_ws_bytes(n_chunks) + 4096);
    if (rc) return rc;
    u8* d_chunks = ar.take(32 * (n_chunks ? n_chunks : 1));
    u8* d_branch = ar.take(32ull * depth);
    u8* ws = ar.take(merkle_ws_bytes(n_chunks));
    if (!d_chunks || !d_branch || !ws) return ECGPU_ERR_OOM;
    if (n_chunks) ECG_HIP_CHECK(hipMemcpyAsync(d_chunks, chunks, 32 * n_chunks, hipMemcpyHostToDevice, s));
    c->last_hash64 = 0;
    rc = proof_chunks_device(s, c, d_chunks, n_chunks, limit, index, d_branch, ws);
    if (rc) return rc;
    ECG_HIP_CHECK(hipMemcpyAsync(branch, d_branch, 32ull * depth, hipMemcpyDeviceToHost, s));
    ECG_HIP_CHECK(hipStreamSynchronize(s));
    return ECGPU_SUCCESS;
}

int ecgpu_ssz_prove(const ecgpu_ssz_type* types, uint32_t n_types, const uint32_t* fields, uint32_t n_field_refs, uint32_t root_type,
                    const uint8_t* ssz, uint64_t n_bytes, const uint64_t* path, uint32_t path_len, uint8_t leaf[32], uint8_t* branch,
                    uint32_t max_depth, uint32_t* depth_out, uint64_t* gindex, uint8_t root[32]) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!types || root_type >= n_types || (!ssz && n_bytes) || (path_len && !path) || !leaf || (!branch && max_depth) || !depth_out ||
        !gindex || !root || (n_field_refs && !fields))
        return ECGPU_ERR_BAD_ARG;
    Walker w{types, n_types, fields, n_field_refs, {}};
    if (!w.valid_types()) {
        set_last_error("bad type description");
        return ECGPU_ERR_BAD_ARG;
    }
    auto bad = [&](const char* m) {
        set_last_error(m);
        return ECGPU_ERR_BAD_ARG;
    };
    // the witness root (also validates the whole encoding against the type)
    rc = ecgpu_htr_ssz(types, n_types, fields, n_field_refs, root_type, ssz, n_bytes, root);
    if (rc) return rc;
    // walk down, collecting per level the sibling nodes TOP-DOWN; reversed at the end
    std::vector<std::vector<u8>> levels;  // each: the branch part of one object, bottom-up
    u64 g = 1;
    u32 ti = root_type;
    const u8* obj = ssz;
    u64 len = n_bytes;
    static const u8 empty[4] = {0, 0, 0, 0};
    if (!obj) obj = empty;
    bool at_leaf_chunk = false;  // the path has landed on a packed chunk / a length node: `leaf` is set
    for (u32 k = 0; k < path_len; k++) {
        if (at_leaf_chunk) return bad("path continues below a leaf");
        const ecgpu_ssz_type& t = types[ti];
        u64 limit;
        bool mix;
        w.tree_shape(ti, limit, mix);
        u32 child;
        bool is_len;
        const u64 elem = path[k];
        if (w.step(ti, elem, g, child, is_len)) {
            set_last_error(w.error);
            return ECGPU_ERR_BAD_ARG;
        }
        // ---- the chunks of this object's data tree (roots of its children, or its packed bytes) and its length
        std::vector<u8> chunks;
        u64 n_chunks = 0, length = 0, pos = 0;
        std::vector<u64> a, b;  // child ranges (the fields of a container)
        bool is_seq = false;    // a vector / list of composite elements: ranges by stride or from the offset table
        u64 seq_fs = 0;
        auto seq_lo = [&](u64 i) -> u64 { return seq_fs != kVariable ? i * seq_fs : (i < n_chunks ? rd32p(obj + 4 * i) : len); };
        if (t.kind == ECGPU_SSZ_CONTAINER) {
            const u32 nf = t.n_fields;
            a.resize(nf);
            b.resize(nf);
            std::vector<u32> var;
            u64 p = 0;
            for (u32 f = 0; f < nf; f++) {
                const u64 fs = w.fixed_size(fields[t.first_field + f]);
                if (fs != kVariable) {
                    a[f] = p;
                    b[f] = p + fs;
                    p += fs;
                } else {
                    if (p + 4 > len) return bad("truncated container");
                    a[f] = rd32p(obj + p);
                    var.push_back(f);
                    p += 4;
                }
            }
            for (size_t v = 0; v < var.size(); v++) b[var[v]] = v + 1 < var.size() ? a[var[v + 1]] : len;
            n_chunks = nf;
            pos = elem;
        } else if ((t.kind == ECGPU_SSZ_VECTOR || t.kind == ECGPU_SSZ_LIST) && types[t.elem].kind != ECGPU_SSZ_UINT) {
            // a homogeneous sequence of composite elements: element ranges on demand (a registry has 2^20 of them)
            seq_fs = w.fixed_size(t.elem);
            u64 cnt = 0;
            if (seq_fs != kVariable) {
                if (seq_fs == 0 || len % seq_fs) return bad("sequence length is not a multiple of the element size");
                cnt = len / seq_fs;
            } else if (len) {
                if (len < 4) return bad("truncated offset table");
                cnt = rd32p(obj) / 4;
                if (cnt == 0 || 4 * cnt > len) return bad("bad first offset");
            }
            is_seq = true;
            n_chunks = cnt;
            length = cnt;
            pos = elem;
        } else {
            // packed bytes: the encoding itself, zero-padded to chunks (bit lists lose their delimiter bit)
            u64 dlen = len;
            chunks.assign(obj, obj + len);
            length = len;
            if (t.kind == ECGPU_SSZ_VECTOR || t.kind == ECGPU_SSZ_LIST) length = len / types[t.elem].param;
            if (t.kind == ECGPU_SSZ_BITVECTOR) length = t.param;
            if (t.kind == ECGPU_SSZ_BITLIST) {
                if (!len || !obj[len - 1]) return bad("bit list without its delimiter");
                u32 msb = 7;
                while (!((obj[len - 1] >> msb) & 1)) msb--;
                length = 8 * (len - 1) + msb;
                chunks[len - 1] &= (u8)((1u << msb) - 1);
                if (msb == 0) dlen = len - 1;
            }
            chunks.resize(((dlen + 31) / 32) * 32, 0);
            n_chunks = chunks.size() / 32;
            pos = g & ((1ull << ceil_log2_u64(limit)) - 1);
        }
        if (t.kind == ECGPU_SSZ_CONTAINER) {
            chunks.assign(32 * (n_chunks ? n_chunks : 1), 0);
            const u32* ftypes = t.kind == ECGPU_SSZ_CONTAINER ? fields + t.first_field : nullptr;
            for (u64 i = 0; i < n_chunks; i++) {
                if (i == pos && !is_len) continue;  // the proven child: its own root is not part of the branch
                if (b[i] < a[i] || b[i] > len) return bad("offsets outside the object");
                rc = ecgpu_htr_ssz(types, n_types, fields, n_field_refs, ftypes ? ftypes[i] : t.elem, obj + a[i], b[i] - a[i], chunks.data() + 32 * i);
                if (rc) return rc;
            }
        }
        // ---- this object's part of the branch
        const u32 depth = ceil_log2_u64(limit);
        std::vector<u8> part(32ull * (depth + (mix ? 1 : 0)), 0);
        u8 data_root[32];
        if (is_len) {
            // proving the length node: its sibling is the root of the data tree
            if (is_seq) {  // merkleize(element roots, limit): the list without its mix-in, one plan
                std::vector<ecgpu_ssz_type> tt(types, types + n_types);
                tt.push_back(ecgpu_ssz_type{ECG_SSZ_LIST_NOMIX, t.elem, limit, 0, 0});
                rc = htr_ssz_impl(tt.data(), n_types + 1, fields, n_field_refs, n_types, obj, len, data_root, true);
            } else {
                rc = ecgpu_merkleize(chunks.data(), 32 * n_chunks, limit, 0, 0, data_root);
            }
            if (rc) return rc;
            std::memcpy(part.data(), data_root, 32);
            part.resize(32);
            std::memset(leaf, 0, 32);
            for (int i = 0; i < 8; i++) leaf[i] = (u8)(length >> (8 * i));
            at_leaf_chunk = true;
        } else {
            if (pos >= (1ull << depth)) return bad("element outside the tree");
            if (depth && is_seq) {
                // The sibling at level l is the root of the aligned subtree over elements [s 2^l, (s + 1) 2^l): ONE plan per level
                // (a list of those elements without its length mix-in: ssz_plan.h ECG_SSZ_LIST_NOMIX), i.e. <= 40 batched
                // launches for state.validators[i] of a 2^20-validator registry.  (Round 2 computed every element's root with a
                // call of its own -- 65 535 host round trips per level -- and refused sequences beyond 65 536 elements.)
                std::vector<ecgpu_ssz_type> tt(types, types + n_types);
                tt.push_back(ecgpu_ssz_type{ECG_SSZ_LIST_NOMIX, t.elem, 0, 0, 0});
                std::vector<u8> synth;
                for (u32 l = 0; l < depth; l++) {
                    const u64 sidx = (pos >> l) ^ 1, lo = sidx << l;
                    u64 hi = (sidx + 1) << l;
                    if (hi > n_chunks) hi = n_chunks;
                    u8* dst = part.data() + 32ull * l;
                    if (lo >= n_chunks) {  // an all-zero subtree of height l: the ladder entry, from the device
                        rc = ecgpu_merkleize(nullptr, 0, 1ull << l, 0, 0, dst);
                        if (rc) return rc;
                        continue;
                    }
                    tt.back().param = 1ull << l;
                    const u64 b0 = seq_lo(lo), b1 = hi < n_chunks ? seq_lo(hi) : len;
                    if (b1 < b0 || b1 > len) return bad("offsets outside the object");
                    if (seq_fs != kVariable) {
                        rc = htr_ssz_impl(tt.data(), n_types + 1, fields, n_field_refs, n_types, obj + b0, b1 - b0, dst, true);
                    } else {
                        // variable-size elements: a list encoding of its own -- a fresh offset table in front of the elements' bytes
                        const u64 k = hi - lo;
                        synth.resize(4 * k + (b1 - b0));
                        for (u64 i = 0; i < k; i++) {
                            const u64 o = 4 * k + (seq_lo(lo + i) - b0);
                            for (int q = 0; q < 4; q++) synth[4 * i + q] = (u8)(o >> (8 * q));
                        }
                        std::memcpy(synth.data() + 4 * k, obj + b0, b1 - b0);
                        rc = htr_ssz_impl(tt.data(), n_types + 1, fields, n_field_refs, n_types, synth.data(), synth.size(), dst, true);
                    }
                    if (rc) return rc;
                }
            } else if (depth) {
                rc = ecgpu_merkle_proof(chunks.data(), n_chunks, limit, pos, part.data());
                if (rc) return rc;
            }
            if (mix) {  // the length mix-in is the top sibling of this object
                u8* top = part.data() + 32ull * depth;
                std::memset(top, 0, 32);
                for (int i = 0; i < 8; i++) top[i] = (u8)(length >> (8 * i));
            }
            if (child == ~0u) {  // landed on a packed chunk
                std::memset(leaf, 0, 32);
                if (pos < n_chunks) std::memcpy(leaf, chunks.data() + 32 * pos, 32);
                at_leaf_chunk = true;
            } else {
                if (pos >= n_chunks) return bad("path selects an element beyond the list's length");
                const u64 c0 = is_seq ? seq_lo(pos) : a[pos], c1 = is_seq ? (pos + 1 < n_chunks ? seq_lo(pos + 1) : len) : b[pos];
                if (c1 < c0 || c1 > len) return bad("offsets outside the object");
                obj = obj + c0;
                len = c1 - c0;
                ti = child;
            }
        }
        levels.push_back(std::move(part));
    }
    if (!at_leaf_chunk) {  // the leaf is the root of the object the path ends at
        rc = ecgpu_htr_ssz(types, n_types, fields, n_field_refs, ti, obj, len, leaf);
        if (rc) return rc;
    }
    u64 total = 0;
    for (auto& p : levels) total += p.size() / 32;
    if (total > max_depth) return bad("branch buffer too small");
    u8* o = branch;
    for (size_t i = levels.size(); i-- > 0;) {
        std::memcpy(o, levels[i].data(), levels[i].size());
        o += levels[i].size();
    }
    *depth_out = (u32)total;
    *gindex = g;
    return ECGPU_SUCCESS;
}

}  // extern "C"
