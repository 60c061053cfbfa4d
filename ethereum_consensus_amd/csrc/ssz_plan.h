// Host-side plan of a generic SSZ hash_tree_root: what ssz_rs's derive macros generate for every container of the
// reference (`#[derive(SimpleSerialize)]`, e.g. /root/reference/ethereum-consensus/src/deneb/beacon_block.rs:12-91),
// driven by a type description (include/ecgpu.h `ecgpu_ssz_type`).  Pure offset arithmetic over the encoding -- no
// hashing, no HIP calls -- so tests/hostsim executes the very same plan on the CPU lane simulator.
//
// Every type instance gets one 32-byte chunk of the "small" buffer for its root.  Basic values are gathered
// straight into their chunk; everything else becomes a finishing job (merkle.h TreeJob: <= 512 nodes -> limit
// depth -> optional mix_in_length) over a contiguous block of child chunks, scheduled at dependency level
// 1 + max(children), or -- above 512 nodes / chunks -- a pass-kernel tree (BigTree) at that level.
#pragma once
#include <string>
#include <vector>

#include "../../include/ecgpu.h"
#include "state_plan.h"

namespace ecg {

// Internal kind (never accepted from a caller: include/ecgpu.h stops at ECGPU_SSZ_CONTAINER): a list WITHOUT its length
// mix-in, i.e. merkleize(element roots, limit).  ecgpu_ssz_prove uses it for the sibling subtrees of a long homogeneous
// sequence: one plan per level instead of one hash_tree_root call per element.
constexpr u32 ECG_SSZ_LIST_NOMIX = 0x100u + ECGPU_SSZ_LIST;


struct SszBigTree {
    LeafKind kind;     // LEAF_CHUNKS: bytes of the encoding; LEAF_NODES: child roots in the small buffer
    u64 src;           // byte offset into the encoding (CHUNKS) or chunk index in the small buffer (NODES)
    u64 bytes, n0;
    u32 depth;
    bool mix;
    u64 mix_len;
    u32 out_chunk;
    u32 level;
};

struct SszPlan {
    std::vector<GatherDesc> gathers;
    std::vector<std::vector<TreeJob>> jobs;  // by dependency level (index 0 unused)
    std::vector<SszBigTree> bigs;
    u32 n_chunks = 1;   // chunk 0 = the root
    u64 hashes = 0;
    std::string error;
};

class SszPlanner {
  public:
    SszPlanner(const ecgpu_ssz_type* types, u32 n_types, const u32* fields, u32 n_field_refs, const u8* enc, u64 n_bytes, SszPlan& plan,
               bool allow_internal = false)
        : T(types), nT(n_types), F(fields), nF(n_field_refs), h(enc), n(n_bytes), P(plan), fixed_memo(n_types, kUnknown), internal_ok(allow_internal) {}

    bool run(u32 root_type) {
        if (!validate()) return false;
        return node(root_type, 0, n, 0, 0) >= 0;
    }

  private:
    static constexpr u64 kUnknown = ~0ull, kVariable = ~0ull - 1;
    const ecgpu_ssz_type* T;
    u32 nT;
    const u32* F;
    u32 nF;
    const u8* h;
    u64 n;
    SszPlan& P;
    std::vector<u64> fixed_memo;
    bool internal_ok;

    int fail(const char* m) {
        if (P.error.empty()) P.error = m;
        return -1;
    }
    bool validate() {
        for (u32 i = 0; i < nT; i++) {
            const ecgpu_ssz_type& t = T[i];
            if (t.kind > ECGPU_SSZ_CONTAINER && !(internal_ok && t.kind == ECG_SSZ_LIST_NOMIX)) return fail("unknown SSZ kind") >= 0;
            if ((t.kind == ECGPU_SSZ_VECTOR || t.kind == ECGPU_SSZ_LIST || t.kind == ECG_SSZ_LIST_NOMIX) && t.elem >= i)
                return fail("element type must precede its container") >= 0;
            if (t.kind == ECGPU_SSZ_CONTAINER) {
                if ((u64)t.first_field + t.n_fields > nF || t.n_fields == 0) return fail("bad field range") >= 0;
                for (u32 k = 0; k < t.n_fields; k++)
                    if (F[t.first_field + k] >= i) return fail("field type must precede its container") >= 0;
            }
            if (t.kind == ECGPU_SSZ_UINT && !(t.param == 1 || t.param == 2 || t.param == 4 || t.param == 8 || t.param == 16 || t.param == 32))
                return fail("bad uint size") >= 0;
            if ((t.kind == ECGPU_SSZ_VECTOR || t.kind == ECGPU_SSZ_BYTEVECTOR || t.kind == ECGPU_SSZ_BITVECTOR) && t.param == 0)
                return fail("empty vector type") >= 0;
        }
        return true;
    }
    // serialized size of a fixed-size type, kVariable otherwise (types are topologically ordered: no recursion depth issue)
    u64 fixed_size(u32 ti) {
        u64& m = fixed_memo[ti];
        if (m != kUnknown) return m;
        const ecgpu_ssz_type& t = T[ti];
        switch (t.kind) {
            case ECGPU_SSZ_UINT:
            case ECGPU_SSZ_BYTEVECTOR: m = t.param; break;
            case ECGPU_SSZ_BITVECTOR: m = (t.param + 7) / 8; break;
            case ECGPU_SSZ_VECTOR: {
                u64 e = fixed_size(t.elem);
                m = e == kVariable ? kVariable : e * t.param;
                break;
            }
            case ECGPU_SSZ_CONTAINER: {
                u64 s = 0;
                for (u32 k = 0; k < t.n_fields && s != kVariable; k++) {
                    u64 e = fixed_size(F[t.first_field + k]);
                    s = e == kVariable ? kVariable : s + e;
                }
                m = s;
                break;
            }
            default: m = kVariable;
        }
        return m;
    }
    bool is_basic(u32 ti) const { return T[ti].kind == ECGPU_SSZ_UINT; }
    u32 alloc(u32 k) {
        u32 r = P.n_chunks;
        P.n_chunks += k;
        return r;
    }
    void job(u32 level, u32 in_chunk, u32 cnt, u32 depth, bool mix, u64 mix_len, u32 out_chunk) {
        if (P.jobs.size() <= level) P.jobs.resize(level + 1);
        TreeJob j;
        j.in_off = 32ull * in_chunk;
        j.out_off = 32ull * out_chunk;
        j.mix_len = mix_len;
        j.n = cnt;
        j.level = 0;
        j.depth = depth;
        j.mix = mix ? 1 : 0;
        P.jobs[level].push_back(j);
        P.hashes += tree_hash_count(LEAF_NODES, cnt, depth, mix);
    }
    static u32 rd32(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }

    // packed bytes [off, off+len) -> chunks -> tree of `limit_chunks` leaves (+ mix-in); returns the level of dst
    int bytes_tree(u64 off, u64 len, u64 limit_chunks, bool mix, u64 mix_len, u32 dst, u32 last_and = 0) {
        const u64 n_chunks = (len + 31) / 32;
        if (limit_chunks == 0) limit_chunks = 1;
        if (n_chunks > limit_chunks) return fail("more chunks than the type's limit");
        const u32 depth = ceil_log2_u64(limit_chunks);
        if (depth == 0 && !mix) {
            if (len) P.gathers.push_back({off, (u32)len, dst, last_and, 0u});
            return 0;
        }
        if (n_chunks <= TREEJOB_MAX_NODES) {
            const u32 blk = alloc((u32)(n_chunks ? n_chunks : 1));
            for (u64 c = 0; c < n_chunks; c++) {
                const u64 nb = (c + 1 == n_chunks) ? len - 32 * c : 32;
                P.gathers.push_back({off + 32 * c, (u32)nb, blk + (u32)c, (c + 1 == n_chunks) ? last_and : 0u, 0u});
            }
            job(1, blk, (u32)n_chunks, depth, mix, mix_len, dst);
            return 1;
        }
        if (last_and) return fail("bit list too long for the generic path");
        P.bigs.push_back({LEAF_CHUNKS, off, len, n_chunks, depth, mix, mix_len, dst, 1u});
        P.hashes += tree_hash_count(LEAF_CHUNKS, n_chunks, depth, mix);
        return 1;
    }

    // roots of `cnt` children already planned into chunks [blk, blk+cnt) at level <= lvl -> tree of `limit` leaves
    int nodes_tree(u32 blk, u64 cnt, int lvl, u64 limit, bool mix, u32 dst) {
        if (limit == 0) limit = 1;
        const u32 depth = ceil_log2_u64(limit);
        if (cnt <= TREEJOB_MAX_NODES) {
            job((u32)lvl + 1, blk, (u32)cnt, depth, mix, cnt, dst);
        } else {
            P.bigs.push_back({LEAF_NODES, blk, 32 * cnt, cnt, depth, mix, cnt, dst, (u32)lvl + 1});
            P.hashes += tree_hash_count(LEAF_NODES, cnt, depth, mix);
        }
        return lvl + 1;
    }

    // plan the instance of type `ti` serialized at [off, off+len); its root goes to chunk `dst`.  Returns the
    // dependency level at which `dst` is valid (0 = gathered), -1 on a malformed encoding.
    int node(u32 ti, u64 off, u64 len, u32 dst, int depth_guard) {
        if (depth_guard > 64) return fail("type nesting too deep");
        if (off > n || len > n - off) return fail("range outside the encoding");
        const ecgpu_ssz_type& t = T[ti];
        switch (t.kind) {
            case ECGPU_SSZ_UINT:
                if (len != t.param) return fail("basic value has the wrong size");
                P.gathers.push_back({off, (u32)len, dst, 0u, 0u});
                return 0;
            case ECGPU_SSZ_BYTEVECTOR:
                if (len != t.param) return fail("byte vector has the wrong size");
                return bytes_tree(off, len, (t.param + 31) / 32, false, 0, dst);
            case ECGPU_SSZ_BITVECTOR:
                if (len != (t.param + 7) / 8) return fail("bit vector has the wrong size");
                return bytes_tree(off, len, (t.param + 255) / 256, false, 0, dst);
            case ECGPU_SSZ_BYTELIST:
                if (len > t.param) return fail("byte list longer than its limit");
                return bytes_tree(off, len, (t.param + 31) / 32, true, len, dst);
            case ECGPU_SSZ_BITLIST: {
                if (len == 0) return fail("bit list without its delimiter byte");
                const u8 last = h[off + len - 1];
                if (last == 0) return fail("bit list without its delimiter bit");
                u32 msb = 7;
                while (!((last >> msb) & 1)) msb--;
                const u64 bits = 8 * (len - 1) + msb;
                if (bits > t.param) return fail("bit list longer than its limit");
                // data = the bits below the delimiter: the last byte loses the delimiter (or disappears entirely)
                const u64 dlen = msb == 0 ? len - 1 : len;
                const u32 mask = msb == 0 ? 0u : (u32)((1u << msb) - 1) | 0x100u;  // 0x100: "mask present" even when it is 0x00..
                return bytes_tree(off, dlen, (t.param + 255) / 256, true, bits, dst, mask);
            }
            case ECGPU_SSZ_VECTOR:
            case ECGPU_SSZ_LIST:
            case ECG_SSZ_LIST_NOMIX: {
                const bool is_list = t.kind != ECGPU_SSZ_VECTOR;
                const bool mix_len = t.kind == ECGPU_SSZ_LIST;
                if (is_basic(t.elem)) {
                    const u64 s = T[t.elem].param;
                    if (len % s) return fail("packed sequence length is not a multiple of the element size");
                    const u64 cnt = len / s;
                    if (is_list ? cnt > t.param : cnt != t.param) return fail("sequence length does not fit the type");
                    return bytes_tree(off, len, (t.param * s + 31) / 32, mix_len, cnt, dst);
                }
                // composite elements: ranges from the stride or from the offset table
                std::vector<u64> starts;
                u64 cnt;
                const u64 fs = fixed_size(t.elem);
                if (fs != kVariable) {
                    if (fs == 0 || len % fs) return fail("sequence length is not a multiple of the element size");
                    cnt = len / fs;
                } else if (len == 0) {
                    cnt = 0;
                } else {
                    if (len < 4) return fail("truncated offset table");
                    const u32 o0 = rd32(h + off);
                    if (o0 % 4 || o0 > len || o0 == 0) return fail("bad first offset");
                    cnt = o0 / 4;
                    starts.resize(cnt + 1);
                    for (u64 i = 0; i < cnt; i++) {
                        starts[i] = rd32(h + off + 4 * i);
                        if (starts[i] > len || (i && starts[i] < starts[i - 1])) return fail("offsets not monotonic");
                    }
                    starts[cnt] = len;
                }
                if (is_list ? cnt > t.param : cnt != t.param) return fail("sequence length does not fit the type");
                if (cnt > 0xfffffffull) return fail("sequence too long");
                const u32 blk = alloc((u32)(cnt ? cnt : 1));
                int lvl = 0;
                for (u64 i = 0; i < cnt; i++) {
                    const u64 a = fs != kVariable ? i * fs : starts[i], b = fs != kVariable ? (i + 1) * fs : starts[i + 1];
                    const int l = node(t.elem, off + a, b - a, blk + (u32)i, depth_guard + 1);
                    if (l < 0) return -1;
                    if (l > lvl) lvl = l;
                }
                return nodes_tree(blk, cnt, lvl, t.param, mix_len, dst);
            }
            default: {  // ECGPU_SSZ_CONTAINER
                const u32 nf = t.n_fields;
                std::vector<u64> a(nf), b(nf);
                std::vector<u32> var;
                u64 pos = 0;
                for (u32 k = 0; k < nf; k++) {
                    const u64 fs = fixed_size(F[t.first_field + k]);
                    if (fs != kVariable) {
                        a[k] = pos;
                        b[k] = pos + fs;
                        pos += fs;
                    } else {
                        if (pos + 4 > len) return fail("truncated container");
                        a[k] = rd32(h + off + pos);
                        var.push_back(k);
                        pos += 4;
                    }
                }
                if (pos > len) return fail("truncated container");
                if (var.empty() ? pos != len : a[var[0]] != pos) return fail("container offsets do not match its fixed part");
                for (size_t v = 0; v < var.size(); v++) {
                    const u64 end = v + 1 < var.size() ? a[var[v + 1]] : len;
                    if (end < a[var[v]] || end > len) return fail("container offsets not monotonic");
                    b[var[v]] = end;
                }
                const u32 blk = alloc(nf);
                int lvl = 0;
                for (u32 k = 0; k < nf; k++) {
                    const int l = node(F[t.first_field + k], off + a[k], b[k] - a[k], blk + k, depth_guard + 1);
                    if (l < 0) return -1;
                    if (l > lvl) lvl = l;
                }
                job((u32)lvl + 1, blk, nf, ceil_log2_u64(nf), false, 0, dst);
                return lvl + 1;
            }
        }
    }
};

inline bool build_ssz_plan(const ecgpu_ssz_type* types, u32 n_types, const u32* fields, u32 n_field_refs, u32 root_type, const u8* enc,
                           u64 n_bytes, SszPlan& plan, bool allow_internal = false) {
    if (!types || root_type >= n_types || (n_field_refs && !fields)) {
        plan.error = "bad type description";
        return false;
    }
    SszPlanner p(types, n_types, fields, n_field_refs, enc, n_bytes, plan, allow_internal);
    return p.run(root_type);
}

}  // namespace ecg
