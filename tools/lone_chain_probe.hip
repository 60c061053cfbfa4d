// Development probe: how fast does ONE lane run a dependent chain of hash64 (the zero ladder of a finishing job) on an otherwise
// idle chip, launch after launch -- and with the rest of the chip busy?  (tools/tail_trace_probe.py: the registry's 29-level
// finishing job takes 12.3 k - 16.8 k cycles per level from one root to the next.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iethereum_consensus_amd/csrc tools/lone_chain_probe.hip -o tools/lone_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "merkle.h"
using namespace ecg;

__global__ void k_chain(u32 n, u32* out, unsigned long long* t) {
    Node x, z;
    for (int i = 0; i < 8; i++) { x.w[i] = threadIdx.x + i; z.w[i] = 0x9e3779b9u * (i + 1); }
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    if (threadIdx.x == 0)
        for (u32 i = 0; i < n; i++) x = hash64(x, z);
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) {
        out[0] = x.w[0];
        t[0] = w1 - w0;
        t[1] = c1 - c0;
        u32 hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        t[2] = hw;
        u32 xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        t[3] = xcc;
    }
}
// throughput waves on every SIMD: all lanes hashing
__global__ void k_busy(u32 n, u32* out) {
    Node x, z;
    for (int i = 0; i < 8; i++) { x.w[i] = threadIdx.x + blockIdx.x + i; z.w[i] = 0x85ebca6bu * (i + 1); }
    for (u32 i = 0; i < n; i++) x = hash64(x, z);
    if (x.w[0] == 0x12345678u) out[1] = x.w[1];
}

int main() {
    u32* d_out;
    unsigned long long* d_t;
    hipMalloc(&d_out, 64);
    hipMalloc(&d_t, 64);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const u32 N = 100;
    for (int mode = 0; mode < 3; mode++) {
        printf("%s\n", mode == 0 ? "idle chip, one launch at a time" : mode == 1 ? "back to back after a chip-filling kernel (same stream)" : "beside a chip-filling kernel (two streams)");
        for (int it = 0; it < 12; it++) {
            if (mode == 1) hipLaunchKernelGGL(k_busy, dim3(4096), dim3(256), 0, s1, 60, d_out);
            if (mode == 2) hipLaunchKernelGGL(k_busy, dim3(4096), dim3(256), 0, s2, 400, d_out);
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s1, N, d_out, d_t);
            hipDeviceSynchronize();
            unsigned long long t[4];
            hipMemcpy(t, d_t, sizeof(t), hipMemcpyDeviceToHost);
            printf("  %5.2f us per hash64, %6.0f counter ticks per hash64 (%4.0f per us), hw_id %08llx xcc %llx\n", t[0] / 100.0 / N, (double)t[1] / N,
                   t[1] / (t[0] / 100.0), t[2], t[3]);
        }
    }
    return 0;
}
