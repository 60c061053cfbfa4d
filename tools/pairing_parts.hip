// Where the lane pairing check spends its time: the Miller loop, the final exponentiation and their building blocks, each
// timed alone at one wave per SIMD (1024 workgroups of 64 lanes, like k_pairing at 65 536 tuples) on register / LDS-resident
// operands.  Development probe (not part of libecgpu.so):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -Iinclude -Iethereum_consensus_amd/csrc \
//         tools/pairing_parts.hip -o tools/pairing_parts && tools/pairing_parts
// Inputs are arbitrary field elements (the arithmetic is data-independent); results are folded into one Fp per lane so that
// nothing is optimised away.
#include <cstdio>
#include <vector>

#include "bls_verify.h"

using namespace ecg;

template <int PART>
__global__ void __launch_bounds__(64, 1) k_part(const Fp* in, Fp* out, int reps) {
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    Fp12 f;
    Fp* fs = (Fp*)&f;
    for (int k = 0; k < 12; k++) fs[k] = in[(t + k) & 63];
    MillerPair pr[2];
    for (int k = 0; k < 2; k++) {
        A1 p;
        A2 q;
        p.x = in[(t + k) & 63];
        p.y = in[(t + k + 1) & 63];
        p.inf = 0;
        q.x = Fp2{in[(t + 2 * k + 2) & 63], in[(t + 3) & 63]};
        q.y = Fp2{in[(t + 4) & 63], in[(t + k + 5) & 63]};
        q.inf = 0;
        miller_pair_init(pr[k], p, q);
    }
    if (PART == 0) miller_loop(f, pr, 2);
    if (PART == 1) {
        Fp12 e;
        final_exponentiation(e, f);
        f = e;
    }
    if (PART == 2)
        for (int i = 0; i < reps; i++) fp12_sqr(f, f);
    if (PART == 3) {
        slot_store_point(0, pr[0].t);
        slot_store_point(1, pr[1].t);
        for (int i = 0; i < reps; i++) miller_dbl_step(f, pr[i & 1], i & 1);
    }
    if (PART == 4)
        for (int i = 0; i < reps; i++) fp12_cyclotomic_sqr_inl(f, f);
    if (PART == 5) {
        slot_store_fp12(f);
        for (int i = 0; i < reps; i++) fp12_mul_by_slots_inl(f, f);
    }
    if (PART == 6) {
        Fp12 g = f;
        for (int i = 0; i < reps; i++) fp12_mul(f, f, g);
    }
    if (PART == 7) {
        Fp12 e;
        fp12_cyc_pow_x(e, f);
        f = e;
    }
    Fp acc = fs[0];
    for (int k = 1; k < 12; k++) acc = fp_add(acc, fs[k]);
    out[t] = acc;
}

template <int PART>
static void run(const char* name, int reps, const Fp* d_in, Fp* d_out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k_part<PART>, dim3(1024), dim3(64), 0, 0, d_in, d_out, reps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k_part<PART>, dim3(1024), dim3(64), 0, 0, d_in, d_out, reps);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("%-46s %8.3f ms", name, ms);
    if (reps > 1) printf("   %8.1f us = %7.0f k cycles per repetition (x %d)", ms * 1e3 / reps, ms * 2.4e3 / reps, reps);
    printf("\n");
}

int main() {
    std::vector<Fp> h(64);
    for (int i = 0; i < 64; i++)
        for (int k = 0; k < 13; k++) h[i].l[k] = (0x12345u * (i + 3) + 0x9e3779u * (k + 1)) & (k == 12 ? 0xfffff : FP_MASK);
    Fp *d_in, *d_out;
    (void)hipMalloc(&d_in, 64 * sizeof(Fp));
    (void)hipMalloc(&d_out, 65536 * sizeof(Fp));
    (void)hipMemcpy(d_in, h.data(), 64 * sizeof(Fp), hipMemcpyHostToDevice);
    run<0>("miller_loop, 2 pairs", 1, d_in, d_out);
    run<1>("final_exponentiation", 1, d_in, d_out);
    run<7>("fp12_cyc_pow_x (one of five)", 1, d_in, d_out);
    run<2>("fp12_sqr (Miller accumulator)", 63, d_in, d_out);
    run<3>("miller_dbl_step incl. line product", 126, d_in, d_out);
    run<4>("fp12_cyclotomic_sqr_inl", 315, d_in, d_out);
    run<5>("fp12_mul_by_slots_inl", 37, d_in, d_out);
    run<6>("fp12_mul (out of line, private segment)", 37, d_in, d_out);
    return 0;
}
