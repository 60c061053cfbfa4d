// Per-opcode issue rates of the integer VALU instructions the two hot paths are made of, measured with instruction streams
// the compiler cannot fold (every instruction is `asm volatile` on eight independent accumulators), at 1 / 2 / 4 / 8 waves per
// SIMD.  This is the table behind the ALU ceilings quoted in DESIGN.md (hash64: v_alignbit_b32 / v_bitop3_b32 / v_add3_u32 /
// v_add_u32; Fp arithmetic: v_mad_u64_u32 / v_mul_lo_u32 / v_add / v_and / v_lshrrev).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/issue_rate.hip -o tools/issue_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned int u32;
typedef unsigned long long u64;

#define REP8(x) x x x x x x x x
#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);      \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

constexpr int TRIPS = 2048;   // loop trips
constexpr int PER_TRIP = 64;  // instructions per trip (8 accumulators x 8)

// 32-bit ops on accumulators a0..a7 with operands b, c
#define OP32(name, line)                                                                                     \
    __global__ void __launch_bounds__(64) k_##name(u32* out) {                                               \
        u32 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        u32 b = blockIdx.x * 2654435761u + 12345u, c = threadIdx.x * 40503u + 7u;                            \
        for (int t = 0; t < TRIPS; t++) {                                                                    \
            REP8(asm volatile(line("%0") "\n\t" line("%1") "\n\t" line("%2") "\n\t" line("%3") "\n\t" line("%4") "\n\t" line("%5") "\n\t" line("%6") "\n\t" line("%7") \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)       \
                              : "v"(b), "v"(c));)                                                            \
        }                                                                                                    \
        out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                          \
    }
#define L_ADD(a) "v_add_u32 " a ", " a ", %8"
#define L_ADD3(a) "v_add3_u32 " a ", " a ", %8, %9"
#define L_XOR(a) "v_xor_b32 " a ", " a ", %8"
#define L_AND(a) "v_and_b32 " a ", " a ", %8"
#define L_BITOP3(a) "v_bitop3_b32 " a ", " a ", %8, %9 bitop3:0x96"
#define L_ALIGNBIT(a) "v_alignbit_b32 " a ", " a ", " a ", 7"
#define L_LSHR(a) "v_lshrrev_b32 " a ", 3, " a
#define L_LSHLADD(a) "v_lshl_add_u32 " a ", " a ", 3, %8"
#define L_MULLO(a) "v_mul_lo_u32 " a ", " a ", %8"
#define L_MULHI(a) "v_mul_hi_u32 " a ", " a ", %8"
#define L_MAD24(a) "v_mad_u32_u24 " a ", " a ", %8, %9"
#define L_PERM(a) "v_perm_b32 " a ", " a ", %8, %9"
#define L_CNDMASK(a) "v_cndmask_b32 " a ", " a ", %8, vcc"
#define L_ADDCO(a) "v_add_co_u32 " a ", vcc, " a ", %8"
OP32(add_u32, L_ADD)
OP32(add3_u32, L_ADD3)
OP32(xor_b32, L_XOR)
OP32(and_b32, L_AND)
OP32(bitop3_b32, L_BITOP3)
OP32(alignbit_b32, L_ALIGNBIT)
OP32(lshrrev_b32, L_LSHR)
OP32(lshl_add_u32, L_LSHLADD)
OP32(mul_lo_u32, L_MULLO)
OP32(mul_hi_u32, L_MULHI)
OP32(mad_u32_u24, L_MAD24)
OP32(perm_b32, L_PERM)

// v_mad_u64_u32 on eight 64-bit accumulators
__global__ void __launch_bounds__(64) k_mad_u64_u32(u32* out) {
    u64 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    u32 b = blockIdx.x * 2654435761u + 12345u, c = threadIdx.x * 40503u + 7u;
    for (int t = 0; t < TRIPS; t++) {
        REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\t"
                          "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                          "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                          : "v"(b), "v"(c)
                          : "vcc");)
    }
    out[blockIdx.x * 64 + threadIdx.x] = (u32)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
// the same with an s_nop 0 after every eight multiply-adds (counted as eight instructions): what the hazard recogniser's s_nop after
// an inline-asm statement costs
__global__ void __launch_bounds__(64) k_mad_u64_u32_nop(u32* out) {
    u64 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    u32 b = blockIdx.x * 2654435761u + 12345u, c = threadIdx.x * 40503u + 7u;
    for (int t = 0; t < TRIPS; t++) {
        REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\t"
                          "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                          "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7\n\ts_nop 0"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                          : "v"(b), "v"(c)
                          : "vcc");)
    }
    out[blockIdx.x * 64 + threadIdx.x] = (u32)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
// the SHA-256 round mix: 4 alignbit + 2 bitop3 + 2 add3 per eight instructions (roughly the hash64 inner loop's blend)
__global__ void __launch_bounds__(64) k_sha_mix(u32* out) {
    u32 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    u32 b = blockIdx.x * 2654435761u + 12345u, c = threadIdx.x * 40503u + 7u;
    for (int t = 0; t < TRIPS; t++) {
        REP8(asm volatile("v_alignbit_b32 %0, %0, %0, 6\n\tv_alignbit_b32 %1, %1, %1, 11\n\tv_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n\t"
                          "v_add3_u32 %3, %3, %8, %9\n\tv_alignbit_b32 %4, %4, %4, 2\n\tv_alignbit_b32 %5, %5, %5, 13\n\t"
                          "v_bitop3_b32 %6, %6, %8, %9 bitop3:0xe8\n\tv_add3_u32 %7, %7, %8, %9"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                          : "v"(b), "v"(c));)
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
// the same eight-accumulator structure with DEPENDENT instructions (one accumulator): the latency a lone chain sees
__global__ void __launch_bounds__(64) k_dep_add3(u32* out) {
    u32 a0 = threadIdx.x;
    u32 b = blockIdx.x * 2654435761u + 12345u, c = threadIdx.x * 40503u + 7u;
    for (int t = 0; t < TRIPS; t++) {
        REP8(asm volatile("v_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %0, %0, %1, %2\n\t"
                          "v_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %0, %0, %1, %2"
                          : "+v"(a0)
                          : "v"(b), "v"(c));)
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0;
}

// round 5, VERDICT item 5(a): would 52-bit limbs on the FP64 unit be cheaper than 30-bit limbs on v_mad_u64_u32?  v_fma_f64 on eight
// independent accumulators; a product of two 52-bit limbs needs TWO of them (the high part, then the low part by an FMA with the
// negated high part) where a product of 30-bit limbs is one v_mad_u64_u32 -- 8 x 8 x 2 = 128 FMAs against 13 x 13 = 169 multiply-adds
// per schoolbook product, so the FMA would have to issue at least as fast as the integer multiply-add to win anything
__global__ void __launch_bounds__(64) k_fma_f64(u32* out) {
    double a0 = threadIdx.x + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = 1.0000001 + blockIdx.x * 1e-9, c = 1e-3 * threadIdx.x;
    for (int t = 0; t < TRIPS; t++) {
        REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                          "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                          : "v"(b), "v"(c));)
    }
    out[blockIdx.x * 64 + threadIdx.x] = (u32)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
// the cross-lane moves of the row machine (csrc/bls_row.h): a DPP row broadcast / row shift per eight, and ds_bpermute
__global__ void __launch_bounds__(64) k_mov_dpp(u32* out) {
    u32 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int t = 0; t < TRIPS; t++) {
        REP8(asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                          "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                          "v_mov_b32_dpp %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %6 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                          "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %0 row_newbcast:12 row_mask:0xf bank_mask:0xf"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
// the inner step of a row sum of products as a DEPENDENT chain (what one iteration of row_sumprod<1> costs a lone wave):
// mad, broadcast of the low dword, mul_lo, and, mad, shift, and, row shift, add -- per eight "instructions" of the table: one step
__global__ void __launch_bounds__(64) k_row_step(u32* out) {
    u64 acc = threadIdx.x;
    const u32 a = threadIdx.x * 2654435761u + 1u, b = blockIdx.x * 40503u + 7u, p = 0x3fffaaabu;
    for (int t = 0; t < TRIPS * 8; t++) {  // the compiler's own sequence for one iteration of csrc/bls_row.h row_sumprod<1>
        acc = (u64)a * b + acc;
        const u32 m = ((u32)__builtin_amdgcn_update_dpp(0, (int)(u32)acc, 0x150, 0xf, 0xf, true) * 0x3ffcfffdu) & 0x3fffffffu;
        acc = (u64)m * p + acc;
        const u64 hi = acc >> 30;
        acc = hi + (u32)__builtin_amdgcn_update_dpp(0, (int)((u32)acc & 0x3fffffffu), 0x101, 0xf, 0xf, true);
        asm volatile("" : "+v"(acc));
    }
    out[blockIdx.x * 64 + threadIdx.x] = (u32)acc;
}

struct Case {
    const char* name;
    void (*fn)(u32*);
};

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int simds = prop.multiProcessorCount * 4;
    const double clk = prop.clockRate * 1e3;
    printf("device %s, %d CUs (%d SIMDs), %.2f GHz; %d instructions per lane per launch\n", prop.gcnArchName, prop.multiProcessorCount, simds,
           clk / 1e9, TRIPS * PER_TRIP);
    printf("%-18s %s\n", "opcode", "cycles per wave-instruction on one SIMD (T lane-ops/s chip-wide) at 1 / 2 / 4 / 8 waves per SIMD");
    u32* d_out;
    CK(hipMalloc(&d_out, (size_t)simds * 8 * 64 * 4));
    const Case cases[] = {{"v_add_u32", k_add_u32},         {"v_add3_u32", k_add3_u32},       {"v_xor_b32", k_xor_b32},
                          {"v_and_b32", k_and_b32},         {"v_bitop3_b32", k_bitop3_b32},   {"v_alignbit_b32", k_alignbit_b32},
                          {"v_lshrrev_b32", k_lshrrev_b32}, {"v_lshl_add_u32", k_lshl_add_u32}, {"v_perm_b32", k_perm_b32},
                          {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mul_lo_u32", k_mul_lo_u32},   {"v_mul_hi_u32", k_mul_hi_u32},
                          {"v_mad_u64_u32", k_mad_u64_u32}, {"8 mad + s_nop", k_mad_u64_u32_nop}, {"sha256 round mix", k_sha_mix},  {"v_add3 dependent", k_dep_add3},
                          {"v_fma_f64", k_fma_f64},         {"v_mov_b32_dpp", k_mov_dpp},     {"row step / 8", k_row_step}};
    for (const Case& c : cases) {
        printf("%-18s", c.name);
        for (int wps = 1; wps <= 8; wps *= 2) {
            const int blocks = simds * wps;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(64), 0, 0, d_out);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(64), 0, 0, d_out);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double instr = (double)TRIPS * PER_TRIP;
            const double cyc = ms * 1e-3 * clk / (instr * wps);  // SIMD cycles per wave-instruction
            const double tops = instr * 64.0 * blocks / (ms * 1e-3) / 1e12;
            printf("  %5.2f (%6.1f T)", cyc, tops);
        }
        printf("\n");
    }
    return 0;
}
