// Where the message stage (k_h2c: RFC 9380 hash_to_curve to G2, one lane per message) spends its time: its parts, each timed
// alone at one wave per SIMD (1024 workgroups of 64 lanes, like k_h2c at 65 536 messages), next to the number of instructions
// the part executes (tools/isa_census.py gives the static counts): a part that takes much more than ~5 cycles per instruction
// is waiting on the private segment.  Development probe (not part of libecgpu.so):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -Iinclude -Iethereum_consensus_amd/csrc \
//         tools/h2c_parts.hip -o tools/h2c_parts && tools/h2c_parts
#include <cstdio>
#include <vector>

#include "bls_verify.h"

using namespace ecg;

template <int PART>
__global__ void __launch_bounds__(64, 1) k_part(const Fp* in, const u8* msgs, Fp* out, int reps) {
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    J2 p, q;
    Fp* ps = (Fp*)&p;
    Fp* qs = (Fp*)&q;
    for (int k = 0; k < 6; k++) {
        ps[k] = in[(t + k) & 63];
        qs[k] = in[(t + 7 * k + 3) & 63];
    }
    Fp acc = in[t & 63];
    if (PART == 0) {
        u8 xm[256];
        for (int i = 0; i < reps; i++) {
            xmd_expand_256(xm, msgs + 32 * (size_t)t, 32);
            acc.l[0] ^= xm[i & 255];
        }
    }
    if (PART == 10) {
        u32 xw[64];
        for (int i = 0; i < reps; i++) {
            xmd_expand_256_msg32(xw, msgs + 32 * (size_t)t);
            acc.l[0] ^= xw[i & 63];
        }
    }
    if (PART == 1)
        for (int i = 0; i < reps; i++) map_to_curve_g2(p, p.x, p.y);
    if (PART == 2)
        for (int i = 0; i < reps; i++) jac_add(p, p, q);
    if (PART == 3)
        for (int i = 0; i < reps; i++) g2_clear_cofactor(p, p);
    if (PART == 4) {
        A2 a;
        for (int i = 0; i < reps; i++) {
            jac_to_aff(a, p);
            p.x = a.x;
            p.y = a.y;
        }
    }
    if (PART == 5) {
        A2 a;
        hash_to_g2(a, msgs + 32 * (size_t)t, 32);
        p.x = a.x;
        p.y = a.y;
    }
    if (PART == 6)
        for (int i = 0; i < reps; i++) jac_mul_xabs(p, p);
    if (PART == 7)
        for (int i = 0; i < reps; i++) acc = fp_pow_pm3d4(acc);
    if (PART == 8)
        for (int i = 0; i < reps; i++) p.x = fp2_inv(p.x);
    if (PART == 9)
        for (int i = 0; i < reps; i++) jac_dbl(p, p);
    for (int k = 0; k < 6; k++) acc = fp_add(acc, ps[k]);
    out[t] = acc;
}

template <int PART>
static void run(const char* name, int reps, const Fp* d_in, const u8* d_msgs, Fp* d_out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k_part<PART>, dim3(1024), dim3(64), 0, 0, d_in, d_msgs, d_out, reps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k_part<PART>, dim3(1024), dim3(64), 0, 0, d_in, d_msgs, d_out, reps);
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
    std::vector<u8> m(32 * 65536);
    for (size_t i = 0; i < m.size(); i++) m[i] = (u8)(i * 2654435761u >> 13);
    Fp *d_in, *d_out;
    u8* d_m;
    (void)hipMalloc(&d_in, 64 * sizeof(Fp));
    (void)hipMalloc(&d_out, 65536 * sizeof(Fp));
    (void)hipMalloc(&d_m, m.size());
    (void)hipMemcpy(d_in, h.data(), 64 * sizeof(Fp), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_m, m.data(), m.size(), hipMemcpyHostToDevice);
    run<5>("hash_to_g2 (the whole stage)", 1, d_in, d_m, d_out);
    run<0>("xmd_expand_256 (19 SHA-256 blocks, bytes)", 8, d_in, d_m, d_out);
    run<10>("xmd_expand_256_msg32 (18 blocks, registers)", 8, d_in, d_m, d_out);
    run<1>("map_to_curve_g2 (x 2 per message)", 4, d_in, d_m, d_out);
    run<7>("  fp_pow_pm3d4 (x 4 per message)", 8, d_in, d_m, d_out);
    run<8>("  fp2_inv (x 2 per message)", 8, d_in, d_m, d_out);
    run<2>("jac_add<Fp2> (x 15 per message)", 32, d_in, d_m, d_out);
    run<9>("jac_dbl<Fp2> out of line", 32, d_in, d_m, d_out);
    run<6>("jac_mul_xabs<Fp2> (x 2 per message)", 2, d_in, d_m, d_out);
    run<3>("g2_clear_cofactor", 2, d_in, d_m, d_out);
    run<4>("jac_to_aff", 4, d_in, d_m, d_out);
    return 0;
}
