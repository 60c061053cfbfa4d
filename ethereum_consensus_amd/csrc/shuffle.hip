// Swap-or-not shuffling on gfx950 (shuffle.h): three launches -- 90 pivots, rounds x ceil(n/256) source blocks,
// one lane per index walking its rounds against the table (11.8 MB for 2^20 indices, L2 / Infinity-Cache resident).
#include <vector>

#include "runtime.h"
#include "shuffle.h"

namespace ecg {

__global__ void k_shuffle_pivots(ShuffleSeed seed, u32 rounds, u64 n, u64* pivots) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rounds) pivots[r] = shuffle_pivot(seed, r, n);
}

__global__ void k_shuffle_sources(ShuffleSeed seed, u32 rounds, u64 n_blocks, u32* table) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (u64)rounds * n_blocks) return;
    u32 d[8];
    shuffle_hash_block(d, seed, (u32)(i / n_blocks), true, (u32)(i % n_blocks));
    u32* o = table + i * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = d[k];
}

// out[i] = in[shuffled_index(i)] (in == nullptr: the permutation itself)
__global__ void k_shuffle_apply(const u64* in, u64 n, u32 rounds, const u64* pivots, const u32* table, u64 n_blocks, u64* out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 j = shuffled_index(i, n, rounds, pivots, table, n_blocks);
    out[i] = in ? in[j] : j;
}

static ShuffleSeed load_seed(const u8 seed[32]) {
    ShuffleSeed s;
    for (int i = 0; i < 8; i++) s.w[i] = ((u32)seed[4 * i] << 24) | ((u32)seed[4 * i + 1] << 16) | ((u32)seed[4 * i + 2] << 8) | seed[4 * i + 3];
    return s;
}

static size_t shuffle_ws_bytes(u64 n, u32 rounds) { return (size_t)rounds * ((n + 255) / 256) * 32 + (size_t)rounds * 8 + 1024; }

static int shuffle_device(hipStream_t s, const u64* d_in, u64 n, const u8 seed[32], u32 rounds, u64* d_out, u8* ws) {
    if (n == 0) return ECGPU_SUCCESS;
    const u64 n_blocks = (n + 255) / 256;
    u64* pivots = (u64*)ws;
    u32* table = (u32*)(ws + (((size_t)rounds * 8 + 255) / 256) * 256);
    const ShuffleSeed sd = load_seed(seed);
    {
        ProfScope ps("shuffle_sources", s);
        if (rounds) hipLaunchKernelGGL(k_shuffle_pivots, dim3((rounds + 63) / 64), dim3(64), 0, s, sd, rounds, n, pivots);
        const u64 nh = (u64)rounds * n_blocks;
        if (nh) hipLaunchKernelGGL(k_shuffle_sources, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, sd, rounds, n_blocks, table);
    }
    {
        ProfScope ps("shuffle_apply", s);
        hipLaunchKernelGGL(k_shuffle_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in, n, rounds, (const u64*)pivots,
                           (const u32*)table, n_blocks, d_out);
    }
    ECG_HIP_CHECK(hipGetLastError());
    return ECGPU_SUCCESS;
}

}  // namespace ecg

using namespace ecg;

extern "C" {

int ecgpu_compute_shuffled_indices_dev(const uint64_t* d_indices, uint64_t n, const uint8_t seed[32], uint32_t rounds, uint64_t* d_out,
                                       ecgpu_stream_t stream) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!seed || (n && !d_out) || rounds > 255) return ECGPU_ERR_BAD_ARG;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(stream);
    Arena& ar = c->arena(s);
    ar.reset();
    rc = ar.reserve(shuffle_ws_bytes(n, rounds) + 1024);
    if (rc) return rc;
    u8* ws = ar.take(shuffle_ws_bytes(n, rounds));
    if (!ws) return ECGPU_ERR_OOM;
    return shuffle_device(s, d_indices, n, seed, rounds, d_out, ws);
}

int ecgpu_compute_shuffled_indices(const uint64_t* indices, uint64_t n, const uint8_t seed[32], uint32_t rounds, uint64_t* out) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!seed || (n && !out) || rounds > 255) return ECGPU_ERR_BAD_ARG;
    if (n == 0) return ECGPU_SUCCESS;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    Arena& ar = c->arena(s);
    ar.reset();
    rc = ar.reserve(shuffle_ws_bytes(n, rounds) + 16 * n + 4096);
    if (rc) return rc;
    u8* ws = ar.take(shuffle_ws_bytes(n, rounds));
    u64* d_in = indices ? (u64*)ar.take(8 * n) : nullptr;
    u64* d_out = (u64*)ar.take(8 * n);
    if (!ws || !d_out || (indices && !d_in)) return ECGPU_ERR_OOM;
    if (indices) ECG_HIP_CHECK(hipMemcpyAsync(d_in, indices, 8 * n, hipMemcpyHostToDevice, s));
    rc = shuffle_device(s, d_in, n, seed, rounds, d_out, ws);
    if (rc) return rc;
    ECG_HIP_CHECK(hipMemcpyAsync(out, d_out, 8 * n, hipMemcpyDeviceToHost, s));
    ECG_HIP_CHECK(hipStreamSynchronize(s));
    return ECGPU_SUCCESS;
}

}  // extern "C"
