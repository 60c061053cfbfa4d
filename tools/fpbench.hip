// Development probe: sustained rate of the field primitives the BLS kernels are built from, as a
// function of resident waves per SIMD.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iethereum_consensus_amd/csrc tools/fpbench.hip -o tools/fpbench
#include <cstdio>
#include <vector>

#include "bls_fp.h"
#include "sha256.h"
using namespace ecg;

constexpr int ITERS = 2000;

// candidate: Fp2 square as ONE out-of-line call with both Montgomery products in one basic block (ILP x2)
typedef u32 fp2_vec26 __attribute__((ext_vector_type(26)));
static __device__ __attribute__((noinline)) fp2_vec26 fp2_sqr_dual_call(fp2_vec26 v) {
    Fp a0, a1;
    for (int i = 0; i < 13; i++) { a0.l[i] = v[i]; a1.l[i] = v[13 + i]; }
    const Fp t0 = fp_mul_body(fp_add_lazy(a0, a1), fp_sub_lazy(a0, a1));
    const Fp t1 = fp_mul_body(a0, a1);
    const Fp d = fp_add(t1, t1);
    fp2_vec26 o;
    for (int i = 0; i < 13; i++) { o[i] = t0.l[i]; o[13 + i] = d.l[i]; }
    return o;
}
// candidate: Fp2 product as ONE call, second operand handed over through a lane-private LDS slot, three products
// in one basic block
__shared__ u32 g_fp2_arg[26 * 64];
static __device__ __attribute__((noinline)) fp2_vec26 fp2_mul_lds_call(fp2_vec26 v) {
    Fp a0, a1, b0, b1;
    for (int i = 0; i < 13; i++) { a0.l[i] = v[i]; a1.l[i] = v[13 + i]; }
    for (int i = 0; i < 13; i++) { b0.l[i] = g_fp2_arg[i * 64 + threadIdx.x]; b1.l[i] = g_fp2_arg[(13 + i) * 64 + threadIdx.x]; }
    const Fp t0 = fp_mul_body(a0, b0);
    const Fp t1 = fp_mul_body(a1, b1);
    const Fp t2 = fp_mul_body(fp_add_lazy(a0, a1), fp_add_lazy(b0, b1));
    const Fp c0 = fp_sub(t0, t1), c1 = fp_sub(fp_sub(t2, t0), t1);
    fp2_vec26 o;
    for (int i = 0; i < 13; i++) { o[i] = c0.l[i]; o[13 + i] = c1.l[i]; }
    return o;
}

template <int OP>
__global__ void __launch_bounds__(64) k_bench(const Fp* in, Fp* out) {
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    Fp x = in[t & 63], y = in[(t + 7) & 63];
    for (int i = 0; i < ITERS; i++) {
        if (OP == 0) x = fp_mul(x, y);                    // out-of-line call
        if (OP == 1) x = fp_mul_body(x, y);               // inlined body
        if (OP == 2) x = fp_sqr(x);
        if (OP == 3) { x = fp_add(x, y); y = fp_sub(y, x); }   // 2 linear ops
        if (OP == 5) {                                     // Fp2 square, current form: 2 calls + inline linear ops
            Fp2 r = fp2_sqr(Fp2{x, y});
            x = r.c0;
            y = r.c1;
        }
        if (OP == 6) {                                     // Fp2 square, one dual-product call
            fp2_vec26 v;
            for (int k = 0; k < 13; k++) { v[k] = x.l[k]; v[13 + k] = y.l[k]; }
            v = fp2_sqr_dual_call(v);
            for (int k = 0; k < 13; k++) { x.l[k] = v[k]; y.l[k] = v[13 + k]; }
        }
        if (OP == 7) {                                     // Fp2 product, current form: 3 calls + inline linear ops
            Fp2 r = fp2_mul(Fp2{x, y}, Fp2{y, x});
            x = r.c0;
            y = r.c1;
        }
        if (OP == 8) {                                     // Fp2 product, one triple-product call, b through LDS
            fp2_vec26 v;
            for (int k = 0; k < 13; k++) { v[k] = x.l[k]; v[13 + k] = y.l[k]; }
            for (int k = 0; k < 13; k++) { g_fp2_arg[k * 64 + threadIdx.x] = y.l[k]; g_fp2_arg[(13 + k) * 64 + threadIdx.x] = x.l[k]; }
            v = fp2_mul_lds_call(v);
            for (int k = 0; k < 13; k++) { x.l[k] = v[k]; y.l[k] = v[13 + k]; }
        }
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
    for (int wps : {1, 2, 4}) {
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
    run<5>("fp2_sqr (2 calls)", 702, 0, d_in, d_out);
    run<6>("fp2_sqr (dual call)", 702, 0, d_in, d_out);
    run<7>("fp2_mul (3 calls)", 1053, 0, d_in, d_out);
    run<8>("fp2_mul (LDS-arg call)", 1053, 0, d_in, d_out);
    return 0;
}
