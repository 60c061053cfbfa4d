// Resident field trees: kernels and host management (see state_tree.h for the algorithm, state_deneb.hip for the caller).
#include "state_tree_host.h"

#include <cstring>

namespace ecg {

// one wave per workgroup: a climb is a dependent chain of hash64 per lane, and 128 waves of dirty entries spread over 128 CUs
// run at the lone-wave rate instead of sharing SIMDs
constexpr int CLIMB_BLOCK = 64;

__global__ void __launch_bounds__(256) k_tree_mark(TreeTable tab, const u64* pairs, u32 n, u64* list, u32* count, u32 cap) {
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const u64 p = pairs[t];
    const u32 slot = (u32)(p >> TREE_SLOT_SHIFT);
    tree_mark(tab.f[slot], slot, p & TREE_ENTRY_MASK, list, count, cap);
}

__global__ void __launch_bounds__(CLIMB_BLOCK) k_tree_climb(TreeTable tab, const u64* list, const u32* count, u32 cap,
                                                            const ZeroTable* zt, unsigned long long* hashes) {
    const u32 t = blockIdx.x * CLIMB_BLOCK + threadIdx.x;
    u32 n = *count;
    if (n > cap) n = cap;
    if (t >= n) return;
    __builtin_amdgcn_s_setprio(3);
    const u64 p = list[t];
    const u32 h = tree_climb(tab.f[(u32)(p >> TREE_SLOT_SHIFT)], p & TREE_ENTRY_MASK, zt);
    if (h) atomicAdd(hashes, (unsigned long long)h);
}

// rebuild: element roots of a record kind
__global__ void __launch_bounds__(256) k_tree_leaves(TreeGeom g) {
    const u64 e = (u64)blockIdx.x * 256 + threadIdx.x;
    if (e >= g.n0) return;
    node_store(tree_leaf(g, e), g.lvl0 + 32ull * e);
}
// rebuild: levels k + 1 .. k + D from level k
template <int D>
__global__ void __launch_bounds__(256) k_tree_build(TreeGeom g, u32 k, const ZeroTable* zt) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= tree_level_count(g.n0, k + D)) return;
    (void)TreeSpan<D>::run(g, k, i, zt);
}

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

void ResidentTrees::release() {
    for (FieldTree& t : f) {
        if (t.block) (void)hipFree(t.block);
        t = FieldTree();
    }
    if (d_list) (void)hipFree(d_list);
    if (d_count) (void)hipFree(d_count);
    d_list = nullptr;
    d_count = nullptr;
    list_cap = bound_total = 0;
    n_slots = 0;
}

int ResidentTrees::sync_geometry(const StatePlan& plan) {
    if (plan.bigs.size() > TREE_MAX_FIELDS) {
        set_last_error("more big fields than tree slots");
        return ECGPU_ERR_BAD_ARG;
    }
    n_slots = (u32)plan.bigs.size();
    u64 want_cap = 1024;
    for (u32 s = 0; s < n_slots; s++) {
        const BigField& b = plan.bigs[s];
        FieldTree& t = f[s];
        const bool cache = b.n0 >= TREE_MIN_ENTRIES && b.kind != LEAF_NODES;
        const u32 H = ceil_log2_u64(b.n0 ? b.n0 : 1);
        if (!cache) {
            if (t.block) {
                ECG_HIP_CHECK(hipDeviceSynchronize());
                ECG_HIP_CHECK(hipFree(t.block));
            }
            const u64 keep_bound = t.bound;  // its stale list entries keep their slots until the next climb skips them
            t = FieldTree();
            t.bound = keep_bound;
            t.g.skip = 1;
            continue;
        }
        if (!t.live || t.g.H != H || t.g.kind != (u32)b.kind) {
            if (t.block) {
                ECG_HIP_CHECK(hipDeviceSynchronize());  // marks of an earlier patch may still be in flight
                ECG_HIP_CHECK(hipFree(t.block));
                t.block = nullptr;
            }
            const u64 cap = 1ull << H;
            const bool records = b.kind != LEAF_CHUNKS;
            const size_t b_lvl0 = records ? up256(32 * cap) : 0, b_nodes = up256(32 * cap), b_cnt = up256(4 * cap),
                         b_flag = up256(4 * ((cap + 31) / 32));
            ECG_HIP_CHECK(hipMalloc((void**)&t.block, b_lvl0 + b_nodes + b_cnt + b_flag));
            t.g.lvl0 = records ? t.block : nullptr;
            t.g.nodes = t.block + b_lvl0;
            t.g.cnt = (u32*)(t.block + b_lvl0 + b_nodes);
            t.g.flag0 = (u32*)(t.block + b_lvl0 + b_nodes + b_cnt);
            t.g.kind = (u32)b.kind;
            t.g.H = H;
            t.g.T = tree_top_level(t.g.kind, H);
            t.live = true;
            t.all_dirty = true;
        }
        t.g.bytes = b.bytes;
        t.g.n0 = b.n0;
        t.src_off = b.src;
        t.out_chunk = b.out_chunk;
        t.depth = b.depth;
        t.mix = b.mix;
        t.mix_len = b.mix_len;
        want_cap += t.share();
    }
    if (!d_count) {
        ECG_HIP_CHECK(hipMalloc((void**)&d_count, 64));
        ECG_HIP_CHECK(hipMemset(d_count, 0, 64));
    }
    if (want_cap > list_cap) {
        u64* nl = nullptr;
        ECG_HIP_CHECK(hipMalloc((void**)&nl, 8 * want_cap));
        if (d_list) {
            ECG_HIP_CHECK(hipDeviceSynchronize());
            ECG_HIP_CHECK(hipMemcpy(nl, d_list, 8 * list_cap, hipMemcpyDeviceToDevice));
            ECG_HIP_CHECK(hipFree(d_list));
        }
        d_list = nl;
        list_cap = want_cap;
    }
    return ECGPU_SUCCESS;
}

void ResidentTrees::collect_entries(u32 slot, u64 first, u64 last, std::vector<u64>& pairs) {
    FieldTree& t = f[slot];
    if (!t.live || t.all_dirty || last < first) return;
    const u64 cnt = last - first + 1;
    if (t.bound + cnt > t.share()) {
        t.all_dirty = true;  // cheaper to rebuild the field's levels in full-width launches
        return;
    }
    t.bound += cnt;
    bound_total += cnt;
    for (u64 e = first; e <= last; e++) pairs.push_back(((u64)slot << TREE_SLOT_SHIFT) | e);
}

void ResidentTrees::collect(u64 lo, u64 hi, std::vector<u64>& pairs) {
    if (hi <= lo) return;
    for (u32 s = 0; s < n_slots; s++) {
        const FieldTree& t = f[s];
        if (!t.live || t.all_dirty) continue;
        const u64 f0 = t.src_off, f1 = t.src_off + t.g.bytes;
        if (hi <= f0 || lo >= f1) continue;
        const u64 rec = leaf_record_bytes((LeafKind)t.g.kind);
        const u64 a = (lo > f0 ? lo : f0) - f0, b = (hi < f1 ? hi : f1) - 1 - f0;
        collect_entries(s, a / rec, b / rec, pairs);
    }
}

static TreeTable make_table(const ResidentTrees& R, const u8* d_ssz) {
    TreeTable tab;
    std::memset(&tab, 0, sizeof(tab));
    for (u32 s = 0; s < TREE_MAX_FIELDS; s++) {
        tab.f[s] = R.f[s].g;
        tab.f[s].src = d_ssz ? d_ssz + R.f[s].src_off : nullptr;
        tab.f[s].skip = (!R.f[s].live || R.f[s].all_dirty) ? 1u : 0u;
    }
    return tab;
}

int ResidentTrees::mark(hipStream_t s, const u64* d_pairs, u32 n) {
    if (!n) return ECGPU_SUCCESS;
    const TreeTable tab = make_table(*this, nullptr);
    hipLaunchKernelGGL(k_tree_mark, dim3((n + 255) / 256), dim3(256), 0, s, tab, d_pairs, n, d_list, d_count, (u32)list_cap);
    ECG_HIP_CHECK(hipGetLastError());
    return ECGPU_SUCCESS;
}

int ResidentTrees::update(hipStream_t s, const u8* d_ssz, u64* hashes) {
    const TreeTable tab = make_table(*this, d_ssz);
    const ZeroTable* zt = device_zero_table();
    for (u32 sl = 0; sl < n_slots; sl++) {
        FieldTree& t = f[sl];
        if (!t.live || !t.all_dirty) continue;
        TreeGeom g = tab.f[sl];
        g.skip = 0;
        const u64 cap = 1ull << g.H;
        // counters and flags of a field that is rebuilt start from zero: marks made before it was flagged are void
        ECG_HIP_CHECK(hipMemsetAsync(g.cnt, 0, 4 * cap + up256(4 * ((cap + 31) / 32)) + (up256(4 * cap) - 4 * cap), s));
        ProfScope ps("merkle_tree_rebuild", s);
        if (g.lvl0) hipLaunchKernelGGL(k_tree_leaves, dim3((unsigned)((g.n0 + 255) / 256)), dim3(256), 0, s, g);
        for (u32 k = 0; k < g.T;) {
            const u32 D = g.T - k >= 3 ? 3 : g.T - k;
            const u64 n_out = tree_level_count(g.n0, k + D);
            const dim3 grid((unsigned)((n_out + 255) / 256));
            if (D == 3) hipLaunchKernelGGL(k_tree_build<3>, grid, dim3(256), 0, s, g, k, zt);
            else if (D == 2) hipLaunchKernelGGL(k_tree_build<2>, grid, dim3(256), 0, s, g, k, zt);
            else hipLaunchKernelGGL(k_tree_build<1>, grid, dim3(256), 0, s, g, k, zt);
            k += D;
        }
        ECG_HIP_CHECK(hipGetLastError());
        if (hashes) *hashes += tree_rebuild_hashes(g);
    }
    if (bound_total) {
        const u64 n = bound_total < list_cap ? bound_total : list_cap;
        ProfScope ps("merkle_tree_climb", s);
        hipLaunchKernelGGL(k_tree_climb, dim3((unsigned)((n + CLIMB_BLOCK - 1) / CLIMB_BLOCK)), dim3(CLIMB_BLOCK), 0, s, tab, (const u64*)d_list,
                           (const u32*)d_count, (u32)list_cap, zt, d_hashes());
        ECG_HIP_CHECK(hipGetLastError());
        ECG_HIP_CHECK(hipMemsetAsync(d_count, 0, 4, s));
    }
    for (u32 sl = 0; sl < TREE_MAX_FIELDS; sl++) {
        f[sl].all_dirty = false;
        f[sl].bound = 0;
    }
    bound_total = 0;
    return ECGPU_SUCCESS;
}

TreeJob ResidentTrees::job(u32 slot, const u8* base, u64 out_off) const {
    const FieldTree& t = f[slot];
    const u8* in = t.g.T == 0 ? t.g.lvl0 : t.g.nodes + 32ull * tree_heap_off(t.g.H, t.g.T);
    TreeJob j;
    j.in_off = (u64)((uintptr_t)in - (uintptr_t)base);  // (modulo 2^64: the trees are allocations of their own)
    j.out_off = out_off;
    j.mix_len = t.mix_len;
    j.n = (u32)tree_level_count(t.g.n0, t.g.T);
    j.level = t.g.T;
    j.depth = t.depth;
    j.mix = t.mix ? 1 : 0;
    return j;
}

u64 ResidentTrees::job_hashes(u32 slot) const {
    const FieldTree& t = f[slot];
    u64 cnt = tree_level_count(t.g.n0, t.g.T), l = t.g.T, h = 0;
    while (cnt > 1) {
        cnt = (cnt + 1) / 2;
        h += cnt;
        l++;
    }
    h += t.depth - l;
    return h + (t.mix ? 1 : 0);
}

}  // namespace ecg
