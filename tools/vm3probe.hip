// Development probe for the lane-group pairing design of round 2 ("vm3"): G lanes share one tuple whose Fp registers live in
// LDS; a ROUND gives every lane one sum of N Fp products (operands = LDS registers picked by a per-lane descriptor, the
// compiled fp_sumprod<N> of csrc/bls_fp.h does the arithmetic) or one lazy linear operation.  The probe runs synthetic rounds
// (random register numbers) and reports multiply-adds per second for each round class at 1 / 2 waves per SIMD, i.e. what
// the interpreter loop + LDS operand traffic cost on top of the arithmetic.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -Iinclude -Iethereum_consensus_amd/csrc \
//         tools/vm3probe.hip -o tools/vm3probe
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bls_fp.h"
using namespace ecg;

constexpr int G = 16;          // lanes per tuple
constexpr int TPW = 64 / G;    // tuples per wave
constexpr int REG_DW = 13;

ECG_D Fp lds_load(const u32* R, u32 r) {
    Fp x;
    const u32* p = R + r * REG_DW;
#pragma unroll
    for (int i = 0; i < 13; i++) x.l[i] = p[i];
    return x;
}
ECG_D void lds_store(u32* R, u32 r, const Fp& x) {
    u32* p = R + r * REG_DW;
#pragma unroll
    for (int i = 0; i < 13; i++) p[i] = x.l[i];
}

template <int N>
ECG_D void round_sumprod(u32* R, const uint4 d) {
    // descriptor bytes: dst, a0..a5, b0..b5 (13 of 16)
    const u32 w[4] = {d.x, d.y, d.z, d.w};
    Fp a[N], b[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int ia = 1 + k, ib = 7 + k;
        a[k] = lds_load(R, (w[ia >> 2] >> ((ia & 3) * 8)) & 255);
        b[k] = lds_load(R, (w[ib >> 2] >> ((ib & 3) * 8)) & 255);
    }
    const Fp r = fp_sumprod<N>(a, b);
    lds_store(R, w[0] & 255, r);
}
ECG_D void round_lin(u32* R, const uint4 d) {
    // d = a + b - c + kp  (lazy: limbs renormalised, no modular correction)
    const Fp a = lds_load(R, (d.x >> 8) & 255), b = lds_load(R, (d.x >> 16) & 255), c = lds_load(R, (d.x >> 24) & 255);
    const Fp k = lds_load(R, d.y & 255);
    Fp s;
    int32_t cy = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        int32_t t = (int32_t)a.l[i] + (int32_t)b.l[i] - (int32_t)c.l[i] + (int32_t)k.l[i] + cy;
        s.l[i] = i + 1 < 13 ? ((u32)t & FP_MASK) : (u32)t;
        cy = t >> 30;
    }
    lds_store(R, d.x & 255, s);
}

// cls: 1, 2, 3, 6 = sum of N products; 0 = linear
__global__ void __launch_bounds__(64) k_vm3(const uint4* prog, const u32* cls, u32 rounds, u32 nreg, const u32* init, u32* out) {
    extern __shared__ u32 lds[];
    const u32 lane = threadIdx.x, slot = lane % G, tl = lane / G;
    u32* R = lds + tl * nreg * REG_DW;
    for (u32 i = slot; i < nreg * REG_DW; i += G) R[i] = init[i] & FP_MASK & (i % 13 == 12 ? 0xfffffu : 0xffffffffu);
    __syncthreads();
    uint4 d = prog[slot];
    u32 c = cls[0];
    for (u32 r = 0; r < rounds; r++) {
        const u32 rn = r + 1 < rounds ? r + 1 : r;
        const uint4 dn = prog[(size_t)rn * G + slot];
        const u32 cn = cls[rn];
        const u32 cu = (u32)__builtin_amdgcn_readfirstlane((int)c);
        if (cu == 6)
            round_sumprod<6>(R, d);
        else if (cu == 3)
            round_sumprod<3>(R, d);
        else if (cu == 2)
            round_sumprod<2>(R, d);
        else if (cu == 1)
            round_sumprod<1>(R, d);
        else
            round_lin(R, d);
        __syncthreads();
        d = dn;
        c = cn;
    }
    u32 acc = 0;
    for (u32 i = slot; i < nreg * REG_DW; i += G) acc ^= R[i];
    out[blockIdx.x * 64 + lane] = acc;
}

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

int main() {
    const u32 nreg = 96;  // 96 x 52 B = 4992 B per tuple, 19 968 B per wave
    const u32 rounds = 400;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    std::vector<u32> init(nreg * REG_DW);
    srand(1);
    for (auto& v : init) v = (u32)rand() * 2654435761u;
    u32 *d_init, *d_out, *d_cls;
    uint4* d_prog;
    CK(hipMalloc(&d_init, init.size() * 4));
    CK(hipMemcpy(d_init, init.data(), init.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_prog, (size_t)rounds * G * 16));
    CK(hipMalloc(&d_cls, rounds * 4));
    CK(hipMalloc(&d_out, (size_t)cus * 64 * 64 * 4));
    struct Mix {
        const char* name;
        std::vector<int> pattern;
        double fill;  // fraction of lanes with real work (all lanes compute anyway; reporting only)
    };
    // "miller": the estimated round mix of one doubling iteration (3 sums of 6, 2 of 1, 2 of 2, 3 linear)
    const Mix mixes[] = {{"S6 only", {6}, 1}, {"S3 only", {3}, 1}, {"S2 only", {2}, 1}, {"S1 only", {1}, 1}, {"LIN only", {0}, 1},
                         {"miller mix 3xS6 2xS2 2xS1 3xLIN", {6, 0, 6, 2, 0, 6, 1, 2, 0, 1}, 1}};
    for (const Mix& m : mixes) {
        std::vector<uint4> prog((size_t)rounds * G);
        std::vector<u32> cls(rounds);
        double mads = 0;
        for (u32 r = 0; r < rounds; r++) {
            const int c = m.pattern[r % m.pattern.size()];
            cls[r] = c;
            if (c) mads += 169.0 * c + 182;
            // destinations of a round are distinct and not read in the same round: dst in [64, 80), sources in [0, 64) U [80, 96)
            for (int s = 0; s < G; s++) {
                u32 b[16];
                b[0] = 64 + s;
                for (int k = 1; k < 16; k++) {
                    u32 v = rand() % 80;
                    b[k] = v < 64 ? v : v + 16;
                }
                // keep values bounded: every few rounds sources are overwritten by results (< 2p), fine for timing
                uint4 d;
                d.x = b[0] | b[1] << 8 | b[2] << 16 | b[3] << 24;
                d.y = b[4] | b[5] << 8 | b[6] << 16 | b[7] << 24;
                d.z = b[8] | b[9] << 8 | b[10] << 16 | b[11] << 24;
                d.w = b[12] | b[13] << 8 | b[14] << 16 | b[15] << 24;
                prog[(size_t)r * G + s] = d;
            }
        }
        CK(hipMemcpy(d_prog, prog.data(), prog.size() * 16, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_cls, cls.data(), cls.size() * 4, hipMemcpyHostToDevice));
        for (int wps = 1; wps <= 2; wps++) {
            // waves per SIMD set by the dynamic LDS size: 160 KB / (4 * wps) per workgroup
            const size_t lds_bytes = wps == 1 ? 40 * 1024 - 64 : 20 * 1024 - 64;
            CK(hipFuncSetAttribute((const void*)k_vm3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            const int blocks = cus * 4 * wps;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(k_vm3, dim3(blocks), dim3(64), lds_bytes, 0, d_prog, d_cls, rounds, nreg, d_init, d_out);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_vm3, dim3(blocks), dim3(64), lds_bytes, 0, d_prog, d_cls, rounds, nreg, d_init, d_out);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double lane_mads = mads * 64.0 * blocks;
            const double cyc_per_round = ms * 1e-3 * prop.clockRate * 1e3 / rounds;
            printf("%-36s %d wave(s)/SIMD: %8.3f ms, %7.0f cycles/round, %6.2f T mads/s\n", m.name, wps, ms, cyc_per_round,
                   lane_mads / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
