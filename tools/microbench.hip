// Integer-issue-rate probe for gfx950: how many lane-ops/s do the instructions the BLS12-381
// limb arithmetic is made of sustain?  (MI355X_MICROARCH.md has no number for v_mad_u64_u32.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;

template <int OP>
__global__ void __launch_bounds__(256) k_probe(uint32_t* out, uint32_t seed) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc[8];
    uint32_t a = t * 2654435761u + seed, b = a ^ 0x9e3779b9u;
    double fa = (double)a, fb = 1.0000001, facc[8];
    for (int i = 0; i < 8; i++) { acc[i] = a + i; facc[i] = a + i; }
    for (int it = 0; it < ITERS / UNROLL; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            int j = u & 7;
            if (OP == 0) acc[j] = (uint64_t)(uint32_t)acc[j] * b + acc[j];                 // v_mad_u64_u32
            if (OP == 1) acc[j] = (uint32_t)acc[j] * b + (uint32_t)(acc[j] >> 32);           // v_mul_lo_u32 (+add)
            if (OP == 2) acc[j] = __umulhi((uint32_t)acc[j], b) + a;                         // v_mul_hi_u32 (+add)
            if (OP == 3) acc[j] = (uint32_t)acc[j] + b + (uint32_t)j;                        // v_add3_u32
            if (OP == 4) facc[j] = __builtin_fma(facc[j], fb, fa);                           // v_fma_f64
            if (OP == 5) acc[j] = __builtin_amdgcn_alignbit((uint32_t)acc[j], b, 7) ^ a;     // alignbit+xor
            if (OP == 6) acc[j] = ((uint32_t)acc[j] & 0xffffff) * (b & 0xffffff) + a;        // v_mad_u32_u24
            if (OP == 7) { uint64_t s = acc[j] + (((uint64_t)a << 32) | b); acc[j] = s; }     // 64-bit add (add_co + addc)
        }
    }
    uint64_t r = 0;
    for (int i = 0; i < 8; i++) r += acc[i] + (uint64_t)facc[i];
    out[t] = (uint32_t)r ^ (uint32_t)(r >> 32);
}

template <int OP>
int run(const char* name, uint32_t* d_out, int blocks) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_probe<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_probe<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 2u + r);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    double ops = 5.0 * blocks * 256.0 * ITERS;
    printf("%-28s %8.3f ms  %8.2f T lane-ops/s\n", name, ms / 5, ops / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    int blocks = 256 * 8 * 4;  // 8 waves/SIMD worth of 256-thread blocks, x4 rounds
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<3>("v_add3_u32", d_out, blocks);
    run<0>("v_mad_u64_u32", d_out, blocks);
    run<1>("v_mul_lo_u32+add", d_out, blocks);
    run<2>("v_mul_hi_u32+add", d_out, blocks);
    run<4>("v_fma_f64", d_out, blocks);
    run<5>("v_alignbit+xor (2 ops)", d_out, blocks);
    run<6>("v_mad_u32_u24 (+2 and)", d_out, blocks);
    run<7>("64-bit add (2 ops)", d_out, blocks);
    return 0;
}
