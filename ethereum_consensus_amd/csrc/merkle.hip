// gfx950 SHA-256 Merkle kernels + host driver + C ABI entry points (see include/ecgpu.h).
#include <thread>
#include <type_traits>
#include <vector>

#include "merkle_driver.h"

#include <cstring>

namespace ecg {

// ---------------------------------------------------------------------------------------------
// device tables
// ---------------------------------------------------------------------------------------------
__device__ ZeroTable g_zero_table;

__global__ void k_init_zero_table() {
    // 64 sequential hash64 on one lane, once per process
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Node z;
    node_zero(z);
    g_zero_table.z[0] = z;
    for (int d = 1; d <= 64; d++) {
        z = hash64(z, z);
        g_zero_table.z[d] = z;
    }
}

static ZeroTable* g_zero_table_ptr[MAX_DEVICES] = {};  // the __device__ table has one instance per device
const ZeroTable* device_zero_table() { return g_zero_table_ptr[current_device()]; }

int init_merkle_tables(hipStream_t s) {
    ECG_HIP_CHECK(hipGetSymbolAddress((void**)&g_zero_table_ptr[current_device()], HIP_SYMBOL(g_zero_table)));
    hipLaunchKernelGGL(k_init_zero_table, dim3(1), dim3(64), 0, s);
    ECG_HIP_CHECK(hipGetLastError());
    return ECGPU_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
constexpr int PASS_BLOCK = 256;  // 4 waves; one lane = one output node

// One pass: n_out nodes, node gid = height-D subtree over level-0 nodes [gid<<D, (gid+1)<<D).
// Consecutive workgroups cover consecutive tiles, so the 8 XCDs stream 8 adjacent 64*2^D-chunk
// tiles at a time (each tile is read once: nothing to share across L2s).
template <int D, class Leaf>
__global__ void __launch_bounds__(PASS_BLOCK) k_merkle_pass(Leaf leaf, u64 n_in, u64 n_out, u8* out,
                                                            const ZeroTable* zt, int level0, u64 gid0) {
    u64 gid = gid0 + (u64)blockIdx.x * PASS_BLOCK + threadIdx.x;
    if (gid >= n_out) return;
    lane_pass<D, Leaf>(leaf, gid, n_in, out, zt, level0);
}

// The registry's leaf pass (D = 2: a lane = four validators = one level-2 node).  In the generic pass a lane reads its own 484
// contiguous bytes, 124 at a time with three root computations in between: neighbouring lanes share every cache line, the
// line's other users come tens of microseconds later, and by then it has left the L1 and the XCD's L2 -- the pass fetched
// 1.39 x the records (PMC, round 4).  Here a WAVE takes its 256 records in four steps of 64 x 121 = 7 744 contiguous bytes:
// 16-byte lane loads into LDS (fully coalesced, every line fetched once and consumed at once), lane i hashes record 64 k + i
// out of LDS, and the 256 roots are transposed through the same LDS so that lane j ends up with records 4 j .. 4 j + 3.
template <>
__global__ void __launch_bounds__(PASS_BLOCK) k_merkle_pass<2, ValidatorLeaves>(ValidatorLeaves leaf, u64 n_in, u64 n_out, u8* out,
                                                                                const ZeroTable* zt, int level0, u64 gid0) {
    __shared__ uint4 stage[PASS_BLOCK / 64][VAL_STAGE_VECS];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 out0 = gid0 + (u64)blockIdx.x * PASS_BLOCK + wave * 64;  // the wave's first output node
    const u64 v0 = out0 << 2;
    if (out0 + 64 > n_out || v0 + 256 > n_in) return;  // (launch_pass sends whole waves only; the ragged rest goes to k_merkle_pass_rest)
    const u8* src = leaf.base + v0 * 121;
    const u32 adj = (u32)((u64)src & 15);  // the same in every step: a step is a multiple of 16 bytes
    const uint4* vsrc = reinterpret_cast<const uint4*>(src - adj);
    uint4* st = stage[wave];
    Node r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint4* g = vsrc + (VAL_STEP_BYTES / 16) * k;
        const uint4 v0_ = g[lane], v1_ = g[lane + 64], v2_ = g[lane + 128], v3_ = g[lane + 192], v4_ = g[lane + 256], v5_ = g[lane + 320],
                    v6_ = g[lane + 384];
        // vectors 448 .. 484: the last one exists only where the step is misaligned (then it holds bytes of the step: mapped memory)
        uint4 v7_ = uint4{0, 0, 0, 0};
        if (lane + 448 < VAL_STAGE_VECS - 1 || (lane + 448 == VAL_STAGE_VECS - 1 && adj)) v7_ = g[lane + 448];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the previous step's reads of the stage are done)
        __builtin_amdgcn_wave_barrier();
        st[lane] = v0_, st[lane + 64] = v1_, st[lane + 128] = v2_, st[lane + 192] = v3_, st[lane + 256] = v4_, st[lane + 320] = v5_,
        st[lane + 384] = v6_;
        if (lane + 448 < VAL_STAGE_VECS) st[lane + 448] = v7_;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        StagedRecord rec = staged_record(reinterpret_cast<const u32*>(st), adj, lane);
        r[k] = validator_root_from_words(rec);
    }
    // roots: lane i holds records 64 k + i; lane j wants 4 j .. 4 j + 3.  Word-major through the stage, four words at a time.
    u32* tw = reinterpret_cast<u32*>(st);
    Node c[4];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int w = 0; w < 4; w++) tw[staged_root_dword(w, 64 * k + lane)] = r[k].w[4 * h + w];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint4 x = *reinterpret_cast<const uint4*>(tw + staged_root_dword(w, 4 * lane));
            c[0].w[4 * h + w] = x.x, c[1].w[4 * h + w] = x.y, c[2].w[4 * h + w] = x.z, c[3].w[4 * h + w] = x.w;
        }
    }
    node_store(hash64(hash64(c[0], c[1]), hash64(c[2], c[3])), out + (out0 + lane) * 32);
}

// the generic pass under a second name: the waves that the registry's staged pass does not take (its ragged end)
template <int D, class Leaf>
__global__ void __launch_bounds__(PASS_BLOCK) k_merkle_pass_rest(Leaf leaf, u64 n_in, u64 n_out, u8* out, const ZeroTable* zt, int level0,
                                                                 u64 gid0) {
    u64 gid = gid0 + (u64)blockIdx.x * PASS_BLOCK + threadIdx.x;
    if (gid >= n_out) return;
    lane_pass<D, Leaf>(leaf, gid, n_in, out, zt, level0);
}

// Finishing jobs: one workgroup per job; level-by-level in LDS, then the zero-ladder climb and
// mix_in_length on lane 0.
constexpr int JOB_BLOCK = 256;
__device__ __forceinline__ void run_tree_job(const TreeJob& job, u8* buf, const ZeroTable* zt) {
    __shared__ Node nodes[TREEJOB_MAX_NODES];
    // the zero ladder in LDS: the climb below is up to ~30 dependent hash64 on one lane, and a global load per step (an L2
    // miss whenever another workgroup's fence has invalidated the cache meanwhile) sat in that chain
    __shared__ Node zl[65];
    const u32 t = threadIdx.x;
    u32 m = job.n;
    for (u32 i = t; i < m; i += JOB_BLOCK) node_load(nodes[i], buf + job.in_off + 32ull * i);
    if (t < 65) zl[t] = zt->z[t];
    __syncthreads();
    u32 lvl = job.level;
    while (m > 1) {
        u32 pairs = (m + 1) >> 1;
        Node h[TREEJOB_MAX_NODES / 2 / JOB_BLOCK];
#pragma unroll
        for (u32 k = 0; k < TREEJOB_MAX_NODES / 2 / JOB_BLOCK; k++) {
            u32 i = t + k * JOB_BLOCK;
            if (i < pairs) {
                Node l = nodes[2 * i];
                Node r = (2 * i + 1 < m) ? nodes[2 * i + 1] : zl[lvl];
                h[k] = hash64(l, r);
            }
        }
        __syncthreads();
#pragma unroll
        for (u32 k = 0; k < TREEJOB_MAX_NODES / 2 / JOB_BLOCK; k++) {
            u32 i = t + k * JOB_BLOCK;
            if (i < pairs) nodes[i] = h[k];
        }
        __syncthreads();
        m = pairs;
        lvl++;
    }
    if (t == 0) {
        Node x = (job.n == 0) ? zl[job.depth] : nodes[0];
        if (job.n != 0) {
            for (; lvl < job.depth; lvl++) x = hash64(x, zl[lvl]);
        }
        if (job.mix) x = hash64(x, len_chunk(job.mix_len));
        node_store(x, buf + job.out_off);
    }
}

// Jobs and tiles are dependent chains of hash64 on a few waves (11 levels in a tile, ~30 in a finishing job); the passes are
// throughput work on every SIMD.  A chain wave that takes turns with three pass waves is three times slower while the pass
// does not notice a few chain waves, so the chain kernels ask for issue priority (s_setprio) and the passes do not.
__global__ void __launch_bounds__(JOB_BLOCK) k_tree_jobs(const TreeJob* jobs, u8* buf, const ZeroTable* zt) {
    __builtin_amdgcn_s_setprio(3);
    TreeJob job = jobs[blockIdx.x];
    run_tree_job(job, buf, zt);
}
__global__ void __launch_bounds__(JOB_BLOCK) k_tree_job1(TreeJob job, u8* buf, const ZeroTable* zt) {
    __builtin_amdgcn_s_setprio(3);
    run_tree_job(job, buf, zt);
}

// Tile stage (merkle.h TileDesc): blockIdx -> (field, tile) by a scan of the few descriptors; the leaf functor is
// selected per workgroup (uniform branch).  Pairs of virtual nodes are ladder entries, not hashed.
template <class Leaf>
__device__ __forceinline__ Node tile_lane_node(const Leaf& leaf, u64 first, u64 n, const ZeroTable* zt, int level0) {
    return Subtree<TILE_D, Leaf>::run(leaf, first, n, zt, level0);
}
__device__ __forceinline__ void run_tile(const TileDesc& d, const ZeroTable* zt) {
    __shared__ Node nodes[TILE_LANES];
    const u32 t = threadIdx.x;
    const u64 tile = blockIdx.x - d.first_wg;
    const u64 first = tile * TILE_NODES + ((u64)t << TILE_D);
    const int l0 = (int)d.level0;
    Node x;
    switch (d.kind) {
        case LEAF_NODES: x = tile_lane_node(NodeLeaves{d.in}, first, d.n0, zt, l0); break;
        case LEAF_BYTES48: x = tile_lane_node(Bytes48Leaves{d.in, d.in_bytes}, first, d.n0, zt, l0); break;
        case LEAF_PAIR64: x = tile_lane_node(Pair64Leaves{d.in, d.in_bytes}, first, d.n0, zt, l0); break;
        case LEAF_ETH1DATA: x = tile_lane_node(Eth1DataLeaves{d.in, d.in_bytes}, first, d.n0, zt, l0); break;
        case LEAF_U64X2: x = tile_lane_node(U64x2Leaves{d.in, d.in_bytes}, first, d.n0, zt, l0); break;
        case LEAF_U64X3: x = tile_lane_node(U64x3Leaves{d.in, d.in_bytes}, first, d.n0, zt, l0); break;
        default: x = tile_lane_node(ChunkLeaves{d.in, d.in_bytes}, first, d.n0, zt, l0); break;
    }
    nodes[t] = x;
    __syncthreads();
    u32 lvl = d.level0 + TILE_D, m = TILE_LANES;
    while (lvl < d.top && m > 1) {
        const u32 pairs = m >> 1;
        Node h;
        if (t < pairs) {
            // left child = node 2t of this level: virtual iff its first level0 node is past the end
            const u64 left_first = tile * TILE_NODES + ((u64)(2 * t) << (lvl - d.level0));
            h = left_first >= d.n0 ? zt->z[lvl + 1] : hash64(nodes[2 * t], nodes[2 * t + 1]);
        }
        __syncthreads();
        if (t < pairs) nodes[t] = h;
        __syncthreads();
        m = pairs;
        lvl++;
    }
    if (t == 0) node_store(nodes[0], d.out + 32ull * tile);
}
__global__ void __launch_bounds__(TILE_LANES) k_tree_tiles(const TileDesc* descs, u32 n_desc, const ZeroTable* zt) {
    u32 f = 0;
    for (u32 i = 1; i < n_desc; i++)
        if (blockIdx.x >= descs[i].first_wg) f = i;
    const TileDesc d = descs[f];
    __builtin_amdgcn_s_setprio(3);  // see k_tree_jobs
    run_tile(d, zt);
}
__global__ void __launch_bounds__(TILE_LANES) k_tree_tiles1(TileDesc d, const ZeroTable* zt) {
    __builtin_amdgcn_s_setprio(3);  // see k_tree_jobs
    run_tile(d, zt);
}

// one gathered chunk: n_bytes of the encoding, zero-padded to the 32 bytes of the chunk
__device__ __forceinline__ void gather_chunk(const u8* src, u64 src_total, const GatherDesc& g, u8* dst) {
    u32 d[8];
    u64 lim = g.src_off + g.n_bytes;
    if (lim > src_total) lim = src_total;
    load_bytes_le<8>(d, src, g.src_off, lim);
    if (g.last_and && g.n_bytes) {
        const u32 b = g.n_bytes - 1;  // byte index inside the chunk
        d[b >> 2] &= ~(0xffu << (8 * (b & 3))) | ((g.last_and & 0xffu) << (8 * (b & 3)));
    }
    u32* q = reinterpret_cast<u32*>(dst + 32ull * g.dst_chunk);
#pragma unroll
    for (int k = 0; k < 8; k++) q[k] = d[k];
}
__global__ void k_gather(const u8* src, u64 src_total, const GatherDesc* desc, u32 n, u8* dst, u64 chk_off, u32 chk_expect, u32* flag) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && flag && chk_off != ~0ull) {
        u32 w[1];
        load_bytes_le<1>(w, src, chk_off, src_total);
        if (w[0] != chk_expect) *flag = 1u;
    }
    if (i >= n) return;
    gather_chunk(src, src_total, desc[i], dst);
}

// The fused tail of a BeaconState root: see merkle_driver.h TailPlan.
#if defined(ECG_TAIL_TRACE)
// development build (tools/build_variant.sh tailtrace "-DECG_TAIL_TRACE" merkle.hip; tools/tail_trace_probe.py): 100 MHz timestamps
// of the tail's milestones: [0] first workgroup starts  [1] first tile of the critical field starts  [2] its last tile arrives
// [3] its finishing job done ([6], [7]: shader clock at [2], [3])  [4] last arrival at the state container  [5] root written
// [8 + f] finishing job of field f done  [40 + j] unit j (no tile stage) done
__device__ unsigned long long g_tail_trace[128];
#define TAIL_TRACE_MIN(i) do { if (threadIdx.x == 0) atomicMin(&g_tail_trace[i], (unsigned long long)wall_clock64()); } while (0)
#define TAIL_TRACE_SET(i) do { __syncthreads(); if (threadIdx.x == 0) g_tail_trace[i] = (unsigned long long)wall_clock64(); } while (0)
#define TAIL_TRACE_CLK(i) do { if (threadIdx.x == 0) g_tail_trace[i] = (unsigned long long)clock64(); } while (0)
#else
#define TAIL_TRACE_MIN(i) ((void)0)
#define TAIL_TRACE_SET(i) ((void)0)
#define TAIL_TRACE_CLK(i) ((void)0)
#endif
__device__ __forceinline__ bool tail_last_arrival(u32* counter, u32 parties) {
    __shared__ u32 ticket;
    __syncthreads();  // the unit's result has been stored (by lane 0) before this
    if (threadIdx.x == 0) {
        __threadfence();  // release: the result is visible device-wide before the ticket is
        ticket = atomicAdd(counter, 1u);
    }
    __syncthreads();
    const bool last = ticket == parties - 1;
    if (last) __threadfence();  // acquire: the other parties' results
    return last;
}
// the gathered chunks among the `n` chunks at byte offset `off` of the job buffer, fetched by the workgroup about to hash them
__device__ __forceinline__ void tail_gather(const TailPlan& P, u8* buf, u64 off, u32 n) {
    // a finishing job's inputs are nodes of a workspace, which lies on EITHER side of the small-chunk buffer (the state driver
    // takes its upload block from the arena first, the workspaces after it)
    if (off < P.small_off || off >= P.small_end) return;
    const u64 c0 = (off - P.small_off) >> 5;
    for (u32 i = threadIdx.x; i < P.n_gathers; i += blockDim.x) {
        const GatherDesc g = P.gathers[i];
        if (g.dst_chunk >= c0 && g.dst_chunk < c0 + n)
            gather_chunk(g.src_sel == 2 ? P.ext2_src : g.src_sel ? P.ext_src : P.src,
                         g.src_sel == 2 ? 32ull * STATE_MAX_FIELD_CHUNKS : g.src_sel ? P.ext_total : P.src_total, g, buf + P.small_off);
    }
    __syncthreads();
}
__global__ void __launch_bounds__(TILE_LANES) k_state_tail(const TailPlan* pl, u8* buf, const ZeroTable* zt) {
    const TailPlan& P = *pl;
    u32 feeds;
    TAIL_TRACE_MIN(0);
    if (blockIdx.x < P.n_tile_wgs) {
        u32 f = 0;
        for (u32 i = 1; i < P.n_fields; i++)
            if (blockIdx.x >= P.fields[i].tile.first_wg) f = i;
        const TileDesc d = P.fields[f].tile;
        // every wave here is a dependent chain of hash64; the critical field's gets issue priority where chains share a SIMD
        if (P.fields[f].prio) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(1);
        if (P.fields[f].prio) TAIL_TRACE_MIN(1);
        run_tile(d, zt);
        if (!tail_last_arrival(&P.counters[f], P.fields[f].n_tiles)) return;
        if (P.fields[f].prio) {
            TAIL_TRACE_SET(2);
            TAIL_TRACE_CLK(6);
        }
        const TreeJob job = P.fields[f].job;
        run_tree_job(job, buf, zt);
        if (P.fields[f].prio) {
            TAIL_TRACE_SET(3);
            TAIL_TRACE_CLK(7);
        }
        TAIL_TRACE_SET(8 + f);
        feeds = P.fields[f].feeds;
    } else {
        const u32 j = blockIdx.x - P.n_tile_wgs;
        const TreeJob job = P.jobs0[j];
        tail_gather(P, buf, job.in_off, job.n);
        run_tree_job(job, buf, zt);
        TAIL_TRACE_SET(40 + j);
        feeds = P.jobs0_feeds[j];
    }
    if (feeds != TAIL_NONE) {  // an input of a nested container: its last input to arrive computes it
        if (!tail_last_arrival(&P.counters[P.n_fields + feeds], P.jobs1_deps[feeds])) return;
        const TreeJob job = P.jobs1[feeds];
        tail_gather(P, buf, job.in_off, job.n);
        run_tree_job(job, buf, zt);
    }
    if (!tail_last_arrival(&P.counters[P.n_fields + P.n_jobs1], P.final_parties)) return;
    TAIL_TRACE_SET(4);
    const TreeJob top = P.job2;
    tail_gather(P, buf, P.froots_off, P.n_froots);  // every chunk slot of the state container: the unused ones are zero descriptors
    run_tree_job(top, buf, zt);
    __syncthreads();
    const u32 t = threadIdx.x;
    bool bad = false;
    if (P.chk_off != ~0ull) {
        u32 w[1];
        load_bytes_le<1>(w, P.src, P.chk_off, P.src_total);
        bad = w[0] != P.chk_expect;
    }
    if (t < 8) reinterpret_cast<u32*>(P.d_root)[t] = bad ? 0xffffffffu : reinterpret_cast<const u32*>(buf + P.root_off)[t];
    if (t == 0 && P.d_status) *P.d_status = bad ? ECGPU_ERR_BAD_ARG : ECGPU_SUCCESS;
    if (P.d_field_roots)  // n_froots x 32 B
        for (u32 k = t; k < 8 * P.n_froots; k += blockDim.x) reinterpret_cast<u32*>(P.d_field_roots)[k] = reinterpret_cast<const u32*>(buf + P.froots_off)[k];
    TAIL_TRACE_SET(5);
}

// crypto::hash: one lane per message (latency path; the batch form is what a caller should use)
__global__ void k_sha256_batch(const u8* data, u64 len, u64 n, u8* out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Sha256Stream s;
    sha256_init(s);
    sha256_update(s, data + i * len, len);
    u32 dg[8];
    sha256_final(s, dg);
    for (int k = 0; k < 8; k++) {
        u32 v = dg[k];
        out[i * 32 + 4 * k + 0] = (u8)(v >> 24);
        out[i * 32 + 4 * k + 1] = (u8)(v >> 16);
        out[i * 32 + 4 * k + 2] = (u8)(v >> 8);
        out[i * 32 + 4 * k + 3] = (u8)v;
    }
}

// Merkle branch fold (is_valid_merkle_branch): one lane, `depth` sequential hash64
__global__ void k_merkle_branch(const u8* leaf_branch_root, u32 depth, u64 index, u8* ok) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Node v, r;
    node_load(v, leaf_branch_root);
    for (u32 i = 0; i < depth; i++) {
        Node b;
        node_load(b, leaf_branch_root + 32ull * (1 + i));
        v = ((index >> i) & 1) ? hash64(b, v) : hash64(v, b);
    }
    node_load(r, leaf_branch_root + 32ull * (1 + depth));
    bool eq = true;
    for (int i = 0; i < 8; i++) eq = eq && (v.w[i] == r.w[i]);
    *ok = eq ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------------
size_t merkle_ws_bytes(u64 n0) {
    // ping-pong node buffers: first pass output <= n0/2 nodes, then geometric
    u64 a = (n0 + 1) / 2 + 64;
    return (size_t)(a * 32 * 2 + 1024);
}

template <class Leaf>
// output nodes [gid0, gid1) of the pass (the whole pass: 0, n_out)
static int launch_pass(hipStream_t s, int D, const Leaf& leaf, u64 n_in, u64 n_out, u8* out, int level0,
                       const char* tag, u64 gid0 = 0, u64 gid1 = ~0ull) {
    if (gid1 > n_out) gid1 = n_out;
    if (gid1 <= gid0) return ECGPU_SUCCESS;
    n_out = gid1;
    dim3 grid((unsigned)((gid1 - gid0 + PASS_BLOCK - 1) / PASS_BLOCK)), block(PASS_BLOCK);
    const ZeroTable* zt = device_zero_table();
    ProfScope ps(tag, s);
    if constexpr (std::is_same<Leaf, ValidatorLeaves>::value) {
        if (D == 2 && gid0 % 64 == 0) {
            // whole waves (64 nodes = 256 records, all present) to the LDS-staged pass, the ragged end to the generic one
            const u64 whole = n_in / 4 < gid1 ? n_in / 4 : gid1;
            const u64 mid = whole > gid0 ? gid0 + (whole - gid0) / 64 * 64 : gid0;
            if (mid > gid0)
                hipLaunchKernelGGL((k_merkle_pass<2, Leaf>), dim3((unsigned)((mid - gid0 + PASS_BLOCK - 1) / PASS_BLOCK)), block, 0, s, leaf, n_in,
                                   mid, out, zt, level0, gid0);
            if (gid1 > mid)
                hipLaunchKernelGGL((k_merkle_pass_rest<2, Leaf>), dim3((unsigned)((gid1 - mid + PASS_BLOCK - 1) / PASS_BLOCK)), block, 0, s, leaf,
                                   n_in, n_out, out, zt, level0, mid);
            ECG_HIP_CHECK(hipGetLastError());
            return ECGPU_SUCCESS;
        }
    }
    switch (D) {
        case 0: hipLaunchKernelGGL((k_merkle_pass<0, Leaf>), grid, block, 0, s, leaf, n_in, n_out, out, zt, level0, gid0); break;
        case 1: hipLaunchKernelGGL((k_merkle_pass<1, Leaf>), grid, block, 0, s, leaf, n_in, n_out, out, zt, level0, gid0); break;
        case 2: hipLaunchKernelGGL((k_merkle_pass<2, Leaf>), grid, block, 0, s, leaf, n_in, n_out, out, zt, level0, gid0); break;
        case 3: hipLaunchKernelGGL((k_merkle_pass<3, Leaf>), grid, block, 0, s, leaf, n_in, n_out, out, zt, level0, gid0); break;
        case 4: hipLaunchKernelGGL((k_merkle_pass<4, Leaf>), grid, block, 0, s, leaf, n_in, n_out, out, zt, level0, gid0); break;
        case 5: hipLaunchKernelGGL((k_merkle_pass<5, Leaf>), grid, block, 0, s, leaf, n_in, n_out, out, zt, level0, gid0); break;
        default: hipLaunchKernelGGL((k_merkle_pass<6, Leaf>), grid, block, 0, s, leaf, n_in, n_out, out, zt, level0, gid0); break;
    }
    ECG_HIP_CHECK(hipGetLastError());
    return ECGPU_SUCCESS;
}

// single-descriptor form of the tile stage: the descriptor travels as a kernel argument
static int launch_tiles_inline(hipStream_t s, const TileDesc& td, u32 n_wg) {
    if (n_wg == 0) return ECGPU_SUCCESS;
    ProfScope ps("merkle_tree_tiles", s);
    hipLaunchKernelGGL(k_tree_tiles1, dim3(n_wg), dim3(TILE_LANES), 0, s, td, device_zero_table());
    ECG_HIP_CHECK(hipGetLastError());
    return ECGPU_SUCCESS;
}

int launch_tiles(hipStream_t s, const TileDesc* d_descs, u32 n_desc, u32 n_wg) {
    if (n_wg == 0 || n_desc == 0) return ECGPU_SUCCESS;
    ProfScope ps("merkle_tree_tiles", s);
    hipLaunchKernelGGL(k_tree_tiles, dim3(n_wg), dim3(TILE_LANES), 0, s, d_descs, n_desc, device_zero_table());
    ECG_HIP_CHECK(hipGetLastError());
    return ECGPU_SUCCESS;
}

int merkleize_device(hipStream_t s, LeafKind kind, const u8* d_in, u64 in_bytes, u64 n0, u32 depth, bool mix,
                     u64 mix_len, u8* d_out, u8* ws, u64* hash_count, TreeJob* deferred, const u8* job_base, bool background,
                     hipEvent_t after_wide_passes, TileDesc* deferred_tile, u32* deferred_tile_wgs, int phase) {
    if (depth > 64) {
        set_last_error("limit too large");
        return ECGPU_ERR_BAD_ARG;
    }
    if (depth < 64 && n0 > (1ull << depth)) {
        set_last_error("more level-0 nodes than the limit allows");
        return ECGPU_ERR_BAD_ARG;
    }
    const MerkleSchedule sc = schedule_merkleize(kind, n0, depth, mix, background);
    const u64 half = ((n0 + 1) / 2 + 64) * 32;
    u8* bufA = ws;
    u8* bufB = ws + half;
    const u8* cur = d_in;
    const bool describe = phase != MERKLEIZE_LAUNCH, launch = phase != MERKLEIZE_DESCRIBE;
    if (phase != MERKLEIZE_ALL && !(deferred && (deferred_tile || !sc.tile))) {
        set_last_error("merkleize_device: a tree described and launched separately needs its tail deferred");
        return ECGPU_ERR_BAD_ARG;
    }
    for (const PassStep& p : sc.passes) {
        u8* out = (cur == bufA) ? bufB : bufA;
        int rc;
        if (!launch) {
            rc = ECGPU_SUCCESS;
        } else if (p.first) {
            // with an `after_wide_passes` event the chip-filling leaf pass goes out in two halves and the event sits
            // between them, so that whatever waits for it overlaps the second half and the tree's tail
            const bool split = after_wide_passes && p.n_out >= (1u << 17);
            const u64 cuts[3] = {0, split ? ((p.n_out / 2 + PASS_BLOCK - 1) / PASS_BLOCK) * PASS_BLOCK : p.n_out, p.n_out};
            rc = ECGPU_SUCCESS;
            for (int h = 0; h < 2 && !rc; h++) {
                const u64 g0 = cuts[h], g1 = cuts[h + 1];
                switch (kind) {
                    case LEAF_CHUNKS: rc = launch_pass(s, p.D, ChunkLeaves{d_in, in_bytes}, p.n_in, p.n_out, out, 0, "merkle_pass_chunks", g0, g1); break;
                    case LEAF_VALIDATORS: rc = launch_pass(s, p.D, ValidatorLeaves{d_in, in_bytes}, p.n_in, p.n_out, out, 0, "merkle_pass_validators", g0, g1); break;
                    case LEAF_BYTES48: rc = launch_pass(s, p.D, Bytes48Leaves{d_in, in_bytes}, p.n_in, p.n_out, out, 0, "merkle_pass_bytes48", g0, g1); break;
                    case LEAF_PAIR64: rc = launch_pass(s, p.D, Pair64Leaves{d_in, in_bytes}, p.n_in, p.n_out, out, 0, "merkle_pass_pair64", g0, g1); break;
                    case LEAF_U64X2: rc = launch_pass(s, p.D, U64x2Leaves{d_in, in_bytes}, p.n_in, p.n_out, out, 0, "merkle_pass_u64x2", g0, g1); break;
                    case LEAF_U64X3: rc = launch_pass(s, p.D, U64x3Leaves{d_in, in_bytes}, p.n_in, p.n_out, out, 0, "merkle_pass_u64x3", g0, g1); break;
                    default: rc = launch_pass(s, p.D, Eth1DataLeaves{d_in, in_bytes}, p.n_in, p.n_out, out, 0, "merkle_pass_eth1data", g0, g1); break;
                }
                if (h == 0 && after_wide_passes && !rc) {
                    ECG_HIP_CHECK(hipEventRecord(after_wide_passes, s));
                    after_wide_passes = nullptr;
                }
            }
        } else {
            rc = launch_pass(s, p.D, NodeLeaves{cur}, p.n_in, p.n_out, out, (int)p.level_in, "merkle_pass_nodes");
        }
        if (rc) return rc;
        cur = out;
    }
    if (after_wide_passes && launch) ECG_HIP_CHECK(hipEventRecord(after_wide_passes, s));
    if (sc.tile) {
        // one workgroup per 1024 nodes
        u8* out = (cur == bufA) ? bufB : bufA;
        TileDesc td;
        td.in = sc.tile_first ? d_in : cur;
        td.in_bytes = sc.tile_first ? in_bytes : 32ull * sc.tile_n_in;
        td.n0 = sc.tile_n_in;
        td.out = out;
        td.kind = sc.tile_first ? (u32)kind : (u32)LEAF_NODES;
        td.level0 = sc.tile_level_in;
        td.top = depth;
        td.first_wg = 0;
        const u32 n_wg = (u32)((sc.tile_n_in + TILE_NODES - 1) / TILE_NODES);
        if (deferred_tile) {
            if (describe) {
                *deferred_tile = td;
                *deferred_tile_wgs = n_wg;
            }
        } else {
            int rc = launch_tiles_inline(s, td, n_wg);
            if (rc) return rc;
        }
        cur = out;
    } else if (deferred_tile_wgs && describe) {
        *deferred_tile_wgs = 0;
    }
    // finishing job: <= 512 nodes at job_level (or the empty tree) -> climb -> mix-in -> d_out
    TreeJob job;
    u8* jbase = const_cast<u8*>(cur);
    job.in_off = 0;
    job.out_off = (u64)((uintptr_t)d_out - (uintptr_t)jbase);
    job.n = sc.job_n;
    job.level = sc.job_level;
    job.depth = depth;
    job.mix = mix ? 1 : 0;
    job.mix_len = mix_len;
    if (deferred) {
        job.in_off = (u64)((uintptr_t)jbase - (uintptr_t)job_base);
        job.out_off = (u64)((uintptr_t)d_out - (uintptr_t)job_base);
        if (describe) *deferred = job;
    } else {
        ProfScope ps("merkle_tree_job", s);
        hipLaunchKernelGGL(k_tree_job1, dim3(1), dim3(JOB_BLOCK), 0, s, job, jbase, device_zero_table());
    }
    ECG_HIP_CHECK(hipGetLastError());
    if (hash_count && describe) *hash_count += sc.hashes;
    return ECGPU_SUCCESS;
}

int launch_tree_jobs(hipStream_t s, const TreeJob* d_jobs, u32 n_jobs, u8* d_buf) {
    if (n_jobs == 0) return ECGPU_SUCCESS;
    ProfScope ps("merkle_tree_jobs", s);
    hipLaunchKernelGGL(k_tree_jobs, dim3(n_jobs), dim3(JOB_BLOCK), 0, s, d_jobs, d_buf, device_zero_table());
    ECG_HIP_CHECK(hipGetLastError());
    return ECGPU_SUCCESS;
}

int launch_gather(hipStream_t s, const u8* d_src, u64 src_total, const GatherDesc* d_desc, u32 n, u8* d_dst, u64 chk_off, u32 chk_expect,
                  u32* d_flag) {
    if (n == 0 && chk_off == ~0ull) return ECGPU_SUCCESS;
    hipLaunchKernelGGL(k_gather, dim3((n + 63) / 64 ? (n + 63) / 64 : 1), dim3(64), 0, s, d_src, src_total, d_desc, n, d_dst, chk_off, chk_expect, d_flag);
    ECG_HIP_CHECK(hipGetLastError());
    return ECGPU_SUCCESS;
}

int launch_state_tail(hipStream_t s, const TailPlan* d_plan, u32 n_wgs, u8* d_buf) {
    if (n_wgs == 0) return ECGPU_ERR_BAD_ARG;
    ProfScope ps("merkle_state_tail", s);
#if defined(ECG_TAIL_TRACE)
    void* tr = nullptr;
    ECG_HIP_CHECK(hipGetSymbolAddress(&tr, HIP_SYMBOL(g_tail_trace)));
    ECG_HIP_CHECK(hipMemsetAsync(tr, 0xff, sizeof(g_tail_trace), s));
#endif
    hipLaunchKernelGGL(k_state_tail, dim3(n_wgs), dim3(TILE_LANES), 0, s, d_plan, d_buf, device_zero_table());
    ECG_HIP_CHECK(hipGetLastError());
    return ECGPU_SUCCESS;
}

// shared body of the host-memory entry points: H2D, run, D2H 32 bytes
static int merkleize_host(LeafKind kind, const u8* h_in, u64 in_bytes, u64 n0, u32 depth, bool mix, u64 mix_len,
                          u8 root[32]) {
    int rc = ensure_init();
    if (rc) return rc;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    Arena& ar = c->arena(s);
    ar.reset();
    size_t need = in_bytes + 256 + merkle_ws_bytes(n0) + 1024;
    rc = ar.reserve(need);
    if (rc) return rc;
    u8* d_in = ar.take(in_bytes + 4);
    u8* ws = ar.take(merkle_ws_bytes(n0));
    u8* d_root = ar.take(32);
    if (in_bytes) ECG_HIP_CHECK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s));
    u64 hc = 0;
    rc = merkleize_device(s, kind, d_in, in_bytes, n0, depth, mix, mix_len, d_root, ws, &hc);
    if (rc) return rc;
    c->last_hash64 = hc;
    ECG_HIP_CHECK(hipMemcpyAsync(root, d_root, 32, hipMemcpyDeviceToHost, s));
    ECG_HIP_CHECK(hipStreamSynchronize(s));
    return ECGPU_SUCCESS;
}

}  // namespace ecg

using namespace ecg;

extern "C" {

#if defined(ECG_TAIL_TRACE)
int ecgpu_debug_tail_trace(uint64_t* out) {  // after a synchronize: the milestones of the last k_state_tail launch
    ECG_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_trace), 128 * sizeof(uint64_t)));
    return ECGPU_SUCCESS;
}
#endif

int ecgpu_merkleize(const uint8_t* data, uint64_t n_bytes, uint64_t limit_chunks, int mix_in_len, uint64_t len,
                    uint8_t root[32]) {
    if ((!data && n_bytes) || !root) return ECGPU_ERR_BAD_ARG;
    u64 n0 = (n_bytes + 31) / 32;
    u64 limit = limit_chunks ? limit_chunks : n0;
    if (n0 > limit) {
        set_last_error("more chunks than limit_chunks");
        return ECGPU_ERR_BAD_ARG;
    }
    return merkleize_host(LEAF_CHUNKS, data, n_bytes, n0, ceil_log2_u64(limit), mix_in_len != 0, len, root);
}

int ecgpu_merkleize_dev(const uint8_t* d_data, uint64_t n_bytes, uint64_t limit_chunks, int mix_in_len,
                        uint64_t len, uint8_t* d_root, ecgpu_stream_t stream) {
    int rc = ensure_init();
    if (rc) return rc;
    u64 n0 = (n_bytes + 31) / 32;
    u64 limit = limit_chunks ? limit_chunks : n0;
    if (n0 > limit) return ECGPU_ERR_BAD_ARG;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(stream);
    Arena& ar = c->arena(s);
    ar.reset();
    rc = ar.reserve(merkle_ws_bytes(n0) + 512);
    if (rc) return rc;
    u8* ws = ar.take(merkle_ws_bytes(n0));
    u64 hc = 0;
    rc = merkleize_device(s, LEAF_CHUNKS, d_data, n_bytes, n0, ceil_log2_u64(limit), mix_in_len != 0, len, d_root,
                          ws, &hc);
    c->last_hash64 = hc;
    return rc;
}

int ecgpu_htr_validators(const uint8_t* ssz121, uint64_t n, uint64_t limit, uint8_t root[32]) {
    if ((!ssz121 && n) || !root || n > limit) return ECGPU_ERR_BAD_ARG;
    return merkleize_host(LEAF_VALIDATORS, ssz121, n * 121, n, ceil_log2_u64(limit), true, n, root);
}

static int validators_tree_dev(const uint8_t* d_ssz121, uint64_t n, uint64_t limit, bool mix, uint8_t* d_root, ecgpu_stream_t stream) {
    int rc = ensure_init();
    if (rc) return rc;
    if (n > limit) return ECGPU_ERR_BAD_ARG;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(stream);
    Arena& ar = c->arena(s);
    ar.reset();
    rc = ar.reserve(merkle_ws_bytes(n) + 512);
    if (rc) return rc;
    u8* ws = ar.take(merkle_ws_bytes(n));
    u64 hc = 0;
    rc = merkleize_device(s, LEAF_VALIDATORS, d_ssz121, n * 121, n, ceil_log2_u64(limit), mix, n, d_root, ws, &hc);
    c->last_hash64 = hc;
    return rc;
}

int ecgpu_htr_validators_dev(const uint8_t* d_ssz121, uint64_t n, uint64_t limit, uint8_t* d_root,
                             ecgpu_stream_t stream) {
    return validators_tree_dev(d_ssz121, n, limit, true, d_root, stream);
}

// One shard's share of a sharded validator list: the root of the aligned subtree of `width` leaves, no length mix-in.
int ecgpu_validators_subtree_root(const uint8_t* ssz121, uint64_t n, uint64_t width, uint8_t root[32]) {
    if ((!ssz121 && n) || !root || n > width || (width & (width - 1)) || !width) return ECGPU_ERR_BAD_ARG;
    return merkleize_host(LEAF_VALIDATORS, ssz121, n * 121, n, ceil_log2_u64(width), false, 0, root);
}

int ecgpu_validators_subtree_root_dev(const uint8_t* d_ssz121, uint64_t n, uint64_t width, uint8_t* d_root,
                                      ecgpu_stream_t stream) {
    if ((width & (width - 1)) || !width) return ECGPU_ERR_BAD_ARG;
    return validators_tree_dev(d_ssz121, n, width, false, d_root, stream);
}

// Top of a sharded list: n_sub sub-roots of aligned `width`-leaf subtrees -> root of the `limit`-leaf tree.  One finishing
// job: the nodes enter at level log2(width), so odd tails pair with the zero hashes of THAT level upwards.
static int subtree_roots_job(hipStream_t s, const u8* d_nodes, u32 n_sub, u64 width, u64 limit, bool mix, u64 len, u8* d_root) {
    if (!width || (width & (width - 1)) || n_sub > TREEJOB_MAX_NODES || limit % width || (u64)n_sub > limit / width) {
        set_last_error("sub-roots: width must be a power of two dividing the limit, at most 512 sub-roots");
        return ECGPU_ERR_BAD_ARG;
    }
    TreeJob job;
    job.in_off = 0;
    job.out_off = (u64)((uintptr_t)d_root - (uintptr_t)d_nodes);
    job.n = n_sub;
    job.level = ceil_log2_u64(width);
    job.depth = ceil_log2_u64(limit);
    job.mix = mix ? 1 : 0;
    job.mix_len = len;
    ProfScope ps("merkle_tree_job", s);
    hipLaunchKernelGGL(k_tree_job1, dim3(1), dim3(JOB_BLOCK), 0, s, job, const_cast<u8*>(d_nodes), device_zero_table());
    ECG_HIP_CHECK(hipGetLastError());
    return ECGPU_SUCCESS;
}

int ecgpu_merkleize_subtree_roots(const uint8_t* sub_roots, uint32_t n_sub, uint64_t width, uint64_t limit, int mix_in_len,
                                  uint64_t len, uint8_t root[32]) {
    if ((!sub_roots && n_sub) || !root) return ECGPU_ERR_BAD_ARG;
    int rc = ensure_init();
    if (rc) return rc;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    Arena& ar = c->arena(s);
    ar.reset();
    rc = ar.reserve(32ull * n_sub + 1024);
    if (rc) return rc;
    u8* d_in = ar.take(32ull * n_sub + 32);
    u8* d_root = ar.take(32);
    if (n_sub) ECG_HIP_CHECK(hipMemcpyAsync(d_in, sub_roots, 32ull * n_sub, hipMemcpyHostToDevice, s));
    rc = subtree_roots_job(s, d_in, n_sub, width, limit, mix_in_len != 0, len, d_root);
    if (rc) return rc;
    ECG_HIP_CHECK(hipMemcpyAsync(root, d_root, 32, hipMemcpyDeviceToHost, s));
    ECG_HIP_CHECK(hipStreamSynchronize(s));
    return ECGPU_SUCCESS;
}

int ecgpu_merkleize_subtree_roots_dev(const uint8_t* d_sub_roots, uint32_t n_sub, uint64_t width, uint64_t limit,
                                      int mix_in_len, uint64_t len, uint8_t* d_root, ecgpu_stream_t stream) {
    int rc = ensure_init();
    if (rc) return rc;
    ThreadCtx* c = tctx();
    return subtree_roots_job(c->stream_or_own(stream), d_sub_roots, n_sub, width, limit, mix_in_len != 0, len, d_root);
}

int ecgpu_htr_beacon_block_header(const uint8_t ssz112[112], uint8_t root[32]) {
    if (!ssz112 || !root) return ECGPU_ERR_BAD_ARG;
    // leaves: slot, proposer_index (u64 -> chunk), parent_root, state_root, body_root -> limit 5 (padded to 8)
    u8 chunks[5 * 32];
    std::memset(chunks, 0, sizeof(chunks));
    std::memcpy(chunks + 0, ssz112 + 0, 8);
    std::memcpy(chunks + 32, ssz112 + 8, 8);
    std::memcpy(chunks + 64, ssz112 + 16, 96);
    return merkleize_host(LEAF_CHUNKS, chunks, sizeof(chunks), 5, 3, false, 0, root);
}

int ecgpu_signing_root(const uint8_t object_root[32], const uint8_t domain[32], uint8_t root[32]) {
    if (!object_root || !domain || !root) return ECGPU_ERR_BAD_ARG;
    u8 chunks[64];
    std::memcpy(chunks, object_root, 32);
    std::memcpy(chunks + 32, domain, 32);
    return merkleize_host(LEAF_CHUNKS, chunks, 64, 2, 1, false, 0, root);
}

int ecgpu_sha256_batch(const uint8_t* data, size_t len, uint64_t n, uint8_t* out) {
    int rc = ensure_init();
    if (rc) return rc;
    if (n == 0) return ECGPU_SUCCESS;
    if ((!data && len) || !out) return ECGPU_ERR_BAD_ARG;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    Arena& ar = c->arena(s);
    ar.reset();
    rc = ar.reserve(len * n + 32 * n + 1024);
    if (rc) return rc;
    u8* d_in = ar.take(len * n + 4);
    u8* d_out = ar.take(32 * n);
    if (len) ECG_HIP_CHECK(hipMemcpyAsync(d_in, data, len * n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sha256_batch, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, d_in, (u64)len, n, d_out);
    ECG_HIP_CHECK(hipGetLastError());
    ECG_HIP_CHECK(hipMemcpyAsync(out, d_out, 32 * n, hipMemcpyDeviceToHost, s));
    ECG_HIP_CHECK(hipStreamSynchronize(s));
    return ECGPU_SUCCESS;
}

int ecgpu_sha256(const uint8_t* data, size_t len, uint8_t out[32]) { return ecgpu_sha256_batch(data, len, 1, out); }

// One big list over several GPUs of ONE process (second row): device g reduces the aligned subtree of W
// validators starting at g W on a host thread bound to it, the 32-byte sub-roots meet in host memory (the exchange step),
// and the first device finishes the top of the tree.
int ecgpu_htr_validators_multi(const int* devices, uint32_t n_devices, const uint8_t* ssz121, uint64_t n, uint64_t limit, uint8_t root[32]) {
    if (!devices || n_devices == 0 || n_devices > (uint32_t)MAX_DEVICES || (!ssz121 && n) || !root || n > limit) return ECGPU_ERR_BAD_ARG;
    // the tree has ceil_log2(limit) levels: a limit that is not a power of two means the next one (as ecgpu_merkleize treats it)
    u64 lim2 = 1;
    while (lim2 < limit) lim2 <<= 1;
    u64 W = 1;
    while (W * n_devices < n) W <<= 1;
    if (W > lim2) W = lim2;  // n <= limit <= lim2: a single subtree then
    const u32 n_sub = n ? (u32)((n + W - 1) / W) : 0;
    std::vector<u8> sub(32ull * (n_sub ? n_sub : 1));
    std::vector<int> rcs;
    std::vector<std::string> errs;
    // persistent per-device workers (runtime.hip): nothing is created or leaked per call
    int rc = run_on_devices(devices, n_sub, [&](unsigned g) -> int {
        const u64 lo = g * W, cnt = n - lo < W ? n - lo : W;
        return ecgpu_validators_subtree_root(ssz121 + 121 * lo, cnt, W, sub.data() + 32ull * g);
    }, rcs, errs);
    if (rc) return rc;
    for (u32 g = 0; g < n_sub; g++)
        if (rcs[g]) {
            set_last_error("device shard " + std::to_string(g) + ": " + errs[g]);
            return rcs[g];
        }
    // the top of the tree on devices[0]'s worker: the caller's own binding is left alone
    rc = run_on_devices(devices, 1, [&](unsigned) -> int { return ecgpu_merkleize_subtree_roots(sub.data(), n_sub, W, lim2, 1, n, root); },
                        rcs, errs);
    if (rc) return rc;
    if (rcs[0]) set_last_error("top of the tree: " + errs[0]);
    return rcs[0];
}

int ecgpu_is_valid_merkle_branch(const uint8_t leaf[32], const uint8_t* branch, uint32_t depth, uint64_t index,
                                 const uint8_t root[32]) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!leaf || (!branch && depth) || !root || depth > 64) return ECGPU_ERR_BAD_ARG;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    Arena& ar = c->arena(s);
    ar.reset();
    size_t nb = 32ull * (depth + 2);
    rc = ar.reserve(nb + 512);
    if (rc) return rc;
    rc = c->staging.reserve(nb + 8);
    if (rc) return rc;
    u8* d = ar.take(nb);
    u8* d_ok = ar.take(8);
    std::memcpy(c->staging.p, leaf, 32);
    if (depth) std::memcpy(c->staging.p + 32, branch, 32ull * depth);
    std::memcpy(c->staging.p + 32ull * (depth + 1), root, 32);
    ECG_HIP_CHECK(hipMemcpyAsync(d, c->staging.p, nb, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_merkle_branch, dim3(1), dim3(64), 0, s, d, depth, index, d_ok);
    ECG_HIP_CHECK(hipGetLastError());
    u8 ok = 0;
    ECG_HIP_CHECK(hipMemcpyAsync(&ok, d_ok, 1, hipMemcpyDeviceToHost, s));
    ECG_HIP_CHECK(hipStreamSynchronize(s));
    return ok ? ECGPU_SUCCESS : ECGPU_VERIFY_FAIL;
}

uint64_t ecgpu_last_hash64_count(void) { return tctx()->last_hash64; }

}  // extern "C"
