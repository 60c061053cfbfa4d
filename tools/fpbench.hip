// Development probe: sustained rate of the field primitives the BLS kernels are built from, as a
// function of resident waves per SIMD.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iethereum_consensus_amd/csrc tools/fpbench.hip -o tools/fpbench
#include <cstdio>
#include <vector>

#include "bls_fp.h"
#include "sha256.h"
using namespace ecg;

constexpr int ITERS = 2000;

template <int OP>
__global__ void __launch_bounds__(64) k_bench(const Fp* in, Fp* out) {
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    Fp x = in[t & 63], y = in[(t + 7) & 63];
    for (int i = 0; i < ITERS; i++) {
        if (OP == 0) x = fp_mul(x, y);                    // out-of-line call
        if (OP == 1) x = fp_mul_body(x, y);               // inlined body
        if (OP == 2) x = fp_sqr(x);
        if (OP == 3) { x = fp_add(x, y); y = fp_sub(y, x); }   // 2 linear ops
        if (OP == 4) {                                     // Fp2 Karatsuba product on (x, y) * (y, x)
            Fp t0 = fp_mul(x, y), t1 = fp_mul(y, x), t2 = fp_mul(fp_add(x, y), fp_add(y, x));
            x = fp_sub(t0, t1);
            y = fp_sub(fp_sub(t2, t0), t1);
        }
    }
    out[t] = fp_add(x, y);
}

// hash64 chain in registers: the unit of the Merkle roofline
__global__ void __launch_bounds__(64) k_hash(const Fp* in, Fp* out) {
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    Node a, b;
    for (int i = 0; i < 8; i++) {
        a.w[i] = in[t & 63].l[i] + t;
        b.w[i] = in[(t + 5) & 63].l[i] ^ t;
    }
    for (int i = 0; i < ITERS / 2; i++) {
        a = hash64(a, b);
        b = hash64(b, a);
    }
    for (int i = 0; i < 8; i++) out[t].l[i] = a.w[i] ^ b.w[i];
}
void run_hash(const Fp* d_in, Fp* d_out) {
    for (int wps : {1, 2, 4, 5, 8}) {
        const int blocks = 256 * 4 * wps;
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        hipLaunchKernelGGL(k_hash, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_hash, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        const double hashes = (double)ITERS * blocks * 64.0;
        printf("hash64 chain           waves/SIMD %d  %8.3f ms  %7.2f G hash64/s  %7.3f us per sequential hash64  (%.0f cycles SIMD-time per wave-hash)\n", wps,
               ms, hashes / (ms * 1e-3) / 1e9, ms * 1e3 / ITERS, ms * 1e3 / ITERS / wps * 2400);
    }
}

template <int OP>
void run(const char* name, double mults_per_iter, double lin_per_iter, const Fp* d_in, Fp* d_out) {
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * 4 * wps;  // one wave per block
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL(k_bench<OP>, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k_bench<OP>, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double wave_ops = (double)ITERS;  // per wave
        const double us_per_op = ms * 1e3 / wave_ops / wps;  // SIMD time per wave-level iteration
        printf("%-22s waves/SIMD %d  %8.3f ms  %7.3f us SIMD-time per wave-iteration  (%.0f cycles @2.4GHz)", name, wps, ms, us_per_op,
               us_per_op * 2400);
        if (mults_per_iter > 0) printf("  %6.2f T mult/s", mults_per_iter * ITERS * blocks * 64.0 / (ms * 1e-3) / 1e12);
        if (lin_per_iter > 0) printf("  %.0f cycles per linear op", us_per_op * 2400 / lin_per_iter);
        printf("\n");
    }
}

int main() {
    std::vector<Fp> h(64);
    for (int i = 0; i < 64; i++)
        for (int k = 0; k < 13; k++) h[i].l[k] = (0x12345u * (i + 3) + 0x9e3779u * (k + 1)) & (k == 12 ? 0xfffff : FP_MASK);
    Fp *d_in, *d_out;
    hipMalloc(&d_in, 64 * sizeof(Fp));
    hipMalloc(&d_out, 256 * 4 * 8 * 64 * sizeof(Fp));
    hipMemcpy(d_in, h.data(), 64 * sizeof(Fp), hipMemcpyHostToDevice);
    run_hash(d_in, d_out);
    run<0>("fp_mul (call)", 351, 0, d_in, d_out);
    run<1>("fp_mul (inline)", 351, 0, d_in, d_out);
    run<2>("fp_sqr (call)", 273, 0, d_in, d_out);
    run<3>("fp_add + fp_sub", 0, 2, d_in, d_out);
    run<4>("fp2 product", 1053, 0, d_in, d_out);
    return 0;
}
