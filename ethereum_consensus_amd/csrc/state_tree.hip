// Resident field trees: kernels and host management (see state_tree.h for the algorithm, state_deneb.hip for the caller).
#include "state_tree_host.h"

#include <cstdlib>
#include <cstring>

namespace ecg {

// ONE WAVE per region: a region holds ~8 dirty entries of a slot's 4 096, its climb is one dependent chain, and the 1 024 active regions of a
// slot are then one wave per SIMD of the chip.  (With 256 threads per region the four waves of a workgroup share a CU and every
// workgroup's ACTIVE wave is its first: four chains per SIMD, 269 us instead of 70 -- profiles/r05c_resident_probe.txt.)
constexpr int CLIMB_BLOCK = 64;

__global__ void __launch_bounds__(256) k_tree_mark(TreeTable tab, const u64* pairs, u32 n, u32* active, u32* count) {
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const u64 p = pairs[t];
    const u32 slot = (u32)(p >> TREE_SLOT_SHIFT);
    tree_mark(tab.f[slot], slot, p & TREE_ENTRY_MASK, active, count);
}

// one workgroup per active region (state_tree.h CLIMB): the region's counters in LDS (4 x 2^T bytes, dynamic)
// development aid (ECGPU_TREE_TRACE=1, tools/resident_probe.py): 100 MHz timestamps per region workgroup: start, lists counted, climbed
__device__ unsigned long long g_tree_trace[3 * 2048];
__global__ void __launch_bounds__(CLIMB_BLOCK) k_tree_climb(TreeTable tab, const u32* active, const u32* count, const ZeroTable* zt,
                                                            unsigned long long* hashes, int trace) {
    extern __shared__ u32 lcnt[];
    __shared__ u32 n_sh;
    u32 n_active = *count;
    if (n_active > TREE_ACTIVE_CAP) n_active = TREE_ACTIVE_CAP;
    if (blockIdx.x >= n_active) return;
    const u32 a = active[blockIdx.x];
    const TreeGeom& g = tab.f[a >> 16];
    const u32 region = a & 0xffffu;
    if (g.skip) return;  // (a field that is rebuilt this time: the rebuild has cleared its lists)
    __builtin_amdgcn_s_setprio(3);
    if (trace && threadIdx.x == 0 && blockIdx.x < 2048) g_tree_trace[3 * blockIdx.x] = wall_clock64();
    if (threadIdx.x == 0) n_sh = g.rcount[region];
    for (u32 i = threadIdx.x; i < (1u << g.T); i += CLIMB_BLOCK) lcnt[i] = 0;
    __syncthreads();
    const u32 n = n_sh;
    const uint16_t* list = g.rlist + ((u64)region << g.T);
    for (u32 j = threadIdx.x; j < n; j += CLIMB_BLOCK) tree_region_count(g, lcnt, list[j]);
    __syncthreads();
    if (threadIdx.x == 0) g.rcount[region] = 0;  // the list is consumed
    if (trace && threadIdx.x == 0 && blockIdx.x < 2048) g_tree_trace[3 * blockIdx.x + 1] = wall_clock64();
    u32 h = 0;
    if (n <= CLIMB_BLOCK) {  // the usual shape: every dirty entry of the region has a lane of its own
        TreePath P;
        if (threadIdx.x < n) tree_region_path(g, lcnt, region, list[threadIdx.x], zt, P);
        __syncthreads();  // every lane has read the counters pass 1 left before any ticket is taken
        if (threadIdx.x < n) h = tree_region_climb(g, lcnt, region, list[threadIdx.x], zt, P);
    } else {  // a crowded region, entries in rounds of 64: a later round's counters have been taken down by the earlier ones, so its
              // entries treat every ancestor as shared (ticket at every level; an ancestor nobody else reaches was counted 1, and
              // whoever takes a counter from 1 to 0 carries on -- the rule of the two-children case covers it)
        for (u32 j = threadIdx.x; j < n; j += CLIMB_BLOCK) {
            TreePath P;
            tree_region_path(g, lcnt, region, list[j], zt, P);
            P.both = 0xffffffffu;
            h += tree_region_climb(g, lcnt, region, list[j], zt, P);
        }
    }
    if (h) atomicAdd(hashes, (unsigned long long)h);
    if (trace && blockIdx.x < 2048) {
        __syncthreads();
        if (threadIdx.x == 0) g_tree_trace[3 * blockIdx.x + 2] = wall_clock64();
    }
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
    if (d_active) (void)hipFree(d_active);
    if (d_count) (void)hipFree(d_count);
    d_active = nullptr;
    d_count = nullptr;
    bound_total = 0;
    n_slots = max_T = 0;
}

int ResidentTrees::sync_geometry(const StatePlan& plan) {
    if (plan.bigs.size() > TREE_MAX_FIELDS) {
        set_last_error("more big fields than tree slots");
        return ECGPU_ERR_BAD_ARG;
    }
    n_slots = (u32)plan.bigs.size();
    max_T = 0;
    for (u32 s = 0; s < n_slots; s++) {
        const BigField& b = plan.bigs[s];
        FieldTree& t = f[s];
        const u32 H = ceil_log2_u64(b.n0 ? b.n0 : 1);
        const bool cache = b.n0 >= TREE_MIN_ENTRIES && b.kind != LEAF_NODES && tree_top_level((u32)b.kind, H) <= TREE_MAX_T;
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
            const u32 T = tree_top_level((u32)b.kind, H);
            const size_t b_lvl0 = records ? up256(32 * cap) : 0, b_nodes = up256(32 * cap), b_flag = up256(4 * ((cap + 31) / 32)),
                         b_rcount = up256(4 * (cap >> T)), b_rlist = up256(2 * cap);
            ECG_HIP_CHECK(hipMalloc((void**)&t.block, b_lvl0 + b_nodes + b_flag + b_rcount + b_rlist));
            t.g.lvl0 = records ? t.block : nullptr;
            t.g.nodes = t.block + b_lvl0;
            t.g.flag0 = (u32*)(t.block + b_lvl0 + b_nodes);  // flag0 and rcount are adjacent: one memset clears both
            t.g.rcount = (u32*)(t.block + b_lvl0 + b_nodes + b_flag);
            t.g.rlist = (uint16_t*)(t.block + b_lvl0 + b_nodes + b_flag + b_rcount);
            t.g.kind = (u32)b.kind;
            t.g.H = H;
            t.g.T = T;
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
        if (t.g.T > max_T) max_T = t.g.T;
    }
    if (!d_count) {
        ECG_HIP_CHECK(hipMalloc((void**)&d_count, 64));
        // hipMemset on device memory is ASYNCHRONOUS with respect to the host and runs on the null stream, which the library's
        // (non-blocking) streams do not wait for: on a busy device the first mark / climb of a new state could meet the counter
        // before the zeros did -- an uninitialised count, garbage list entries, a memory violation in k_tree_climb (found by the
        // round-6 soak: one fault per ~400 states created beside 16 verifying threads; tests/_soak.py, HISTORY.md round 6).
        ECG_HIP_CHECK(hipMemsetAsync(d_count, 0, 64, nullptr));
        ECG_HIP_CHECK(hipStreamSynchronize(nullptr));
    }
    if (!d_active) ECG_HIP_CHECK(hipMalloc((void**)&d_active, 4 * TREE_ACTIVE_CAP));
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
    for (u64 e = first; e <= last; e++) {
        pairs.push_back(((u64)slot << TREE_SLOT_SHIFT) | e);
        const u64 r = e >> t.g.T;
        if (!(t.region_bits[r >> 6] >> (r & 63) & 1)) {
            t.region_bits[r >> 6] |= 1ull << (r & 63);
            active_regions++;
        }
    }
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
    hipLaunchKernelGGL(k_tree_mark, dim3((n + 255) / 256), dim3(256), 0, s, tab, d_pairs, n, d_active, d_count);
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
        // flags and region lists of a field that is rebuilt start from zero: marks made before it was flagged are void
        ECG_HIP_CHECK(hipMemsetAsync(g.flag0, 0, up256(4 * ((cap + 31) / 32)) + up256(4 * (cap >> g.T)), s));
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
    if (active_regions) {
        // one single-wave workgroup per active region, and LDS claimed so that a CU takes four of them -- one per SIMD: a
        // climb is a dependent chain of hash64, and chains that share a SIMD take turns (at 8 KB of LDS the dispatcher packed
        // the 1 024 regions of a slot several to a SIMD: 25 us per level instead of 6, profiles/r05e_resident_probe_trace.txt)
        const u32 n = active_regions < TREE_ACTIVE_CAP ? active_regions : TREE_ACTIVE_CAP;
        // (36 KB + the few static bytes: four fit a CU's 160 KB, a fifth does not; at 40 KB only three did and a quarter of a
        // slot's regions started 70 us late)
        static const u32 claim_kb = [] { const char* e = getenv("ECGPU_TREE_LDS_KB"); return e ? (u32)atoi(e) : 36u; }();
        const u32 lds = (4u << max_T) > claim_kb * 1024 ? (4u << max_T) : claim_kb * 1024;
        ProfScope ps("merkle_tree_climb", s);
        static const int trace = [] { const char* e = getenv("ECGPU_TREE_TRACE"); return e ? atoi(e) : 0; }();
        if (trace) {
            void* tr = nullptr;
            ECG_HIP_CHECK(hipGetSymbolAddress(&tr, HIP_SYMBOL(g_tree_trace)));
            ECG_HIP_CHECK(hipMemsetAsync(tr, 0, sizeof(g_tree_trace), s));
        }
        hipLaunchKernelGGL(k_tree_climb, dim3(n), dim3(CLIMB_BLOCK), lds, s, tab, (const u32*)d_active, (const u32*)d_count, zt,
                           d_hashes(), trace);
        ECG_HIP_CHECK(hipGetLastError());
        ECG_HIP_CHECK(hipMemsetAsync(d_count, 0, 4, s));
    }
    for (u32 sl = 0; sl < TREE_MAX_FIELDS; sl++) {
        f[sl].all_dirty = false;
        f[sl].bound = 0;
        std::memset(f[sl].region_bits, 0, sizeof(f[sl].region_bits));
    }
    bound_total = 0;
    active_regions = 0;
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

extern "C" int ecgpu_debug_tree_trace(unsigned long long* out /* 3 x 2048 */) {
    ECG_HIP_CHECK(hipDeviceSynchronize());
    ECG_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(ecg::g_tree_trace), sizeof(ecg::g_tree_trace)));
    return ECGPU_SUCCESS;
}
