// Does a wave instruction cost less when whole 16-lane rows of EXEC are off?  v_mad_u64_u32 and v_and_b32 streams (asm volatile,
// eight independent accumulators) with (a) the first 16 / 32 / 48 / 64 lanes active (whole rows off), (b) 4 / 8 / 12 / 16 lanes of
// EVERY row active (no row off), at 1 and 8 waves per SIMD.  If (a) scales with the rows and (b) does not, a lane-group layout that
// gathers idle slots into whole rows turns idle lanes into time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exec_rows.hip -o tools/exec_rows
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned int u32;
typedef unsigned long long u64;
#define REP8(x) x x x x x x x x
#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)
constexpr int TRIPS = 2048;

__global__ void __launch_bounds__(64) k_mad(u32* out, u32 mode, u32 k) {
    const u32 lane = threadIdx.x;
    const bool on = mode == 0 ? lane < k : (lane & 15) < k;
    if (!on) return;
    u64 a0 = lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    u32 b = blockIdx.x * 2654435761u + 12345u, c = lane * 40503u + 7u;
    for (int t = 0; t < TRIPS; t++) {
        REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\t"
                          "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                          "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                          : "v"(b), "v"(c)
                          : "vcc");)
    }
    out[blockIdx.x * 64 + lane] = (u32)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void __launch_bounds__(64) k_and(u32* out, u32 mode, u32 k) {
    const u32 lane = threadIdx.x;
    const bool on = mode == 0 ? lane < k : (lane & 15) < k;
    if (!on) return;
    u32 a0 = lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    u32 b = blockIdx.x * 2654435761u + 12345u;
    for (int t = 0; t < TRIPS; t++) {
        REP8(asm volatile("v_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\t"
                          "v_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                          : "v"(b));)
    }
    out[blockIdx.x * 64 + lane] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int simds = prop.multiProcessorCount * 4;
    const double clk = prop.clockRate * 1e3;
    u32* d_out;
    CK(hipMalloc(&d_out, (size_t)simds * 8 * 64 * 4));
    printf("cycles per wave-instruction on one SIMD at 1 / 8 waves per SIMD\n");
    for (int op = 0; op < 2; op++)
        for (u32 mode = 0; mode < 2; mode++)
            for (u32 q = 1; q <= 4; q++) {
                const u32 k = mode == 0 ? 16 * q : 4 * q;
                printf("%-14s %-34s", op == 0 ? "v_mad_u64_u32" : "v_and_b32", mode == 0 ? "first k lanes (whole rows off), k =" : "k lanes of every row, k =");
                printf(" %2u:", k);
                for (int wps = 1; wps <= 8; wps *= 8) {
                    const int blocks = simds * wps;
                    hipEvent_t e0, e1;
                    CK(hipEventCreate(&e0));
                    CK(hipEventCreate(&e1));
                    for (int rep = 0; rep < 2; rep++) {
                        if (rep == 1) CK(hipEventRecord(e0));
                        if (op == 0)
                            hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(64), 0, 0, d_out, mode, k);
                        else
                            hipLaunchKernelGGL(k_and, dim3(blocks), dim3(64), 0, 0, d_out, mode, k);
                        if (rep == 0) CK(hipDeviceSynchronize());
                    }
                    CK(hipEventRecord(e1));
                    CK(hipDeviceSynchronize());
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    printf("  %5.2f", ms * 1e-3 * clk / ((double)TRIPS * 64 * wps));
                }
                printf("\n");
            }
    return 0;
}
