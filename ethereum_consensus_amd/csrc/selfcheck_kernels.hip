// Probe kernels of the box self-check (selfcheck.hip has the why and the host side): the same multiply-adds as loops over
// 8 KB, 64 KB, 256 KB and 1 MB of straight-line code, one wave per SIMD like the BLS lane kernels.
#include <hip/hip_runtime.h>

#include "common.h"

namespace ecg {

// MADS independent-accumulator multiply-adds per loop trip, emitted as instructions (nothing for the optimizer to fold)
template <int MADS>
__global__ void __launch_bounds__(64) k_ifetch_probe(u32* out, u32 trips) {
    u64 acc0 = threadIdx.x, acc1 = blockIdx.x, acc2 = 3, acc3 = 5;
    const u32 a = threadIdx.x * 2654435761u + 1, b = blockIdx.x * 40503u + 7;
    for (u32 t = 0; t < trips; t++) {
#pragma unroll
        for (int k = 0; k < MADS / 4; k++) {
            asm volatile(
                "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
                "v_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3)
                : "v"(a), "v"(b)
                : "vcc");
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = (u32)(acc0 ^ acc1 ^ acc2 ^ acc3);
}

void launch_ifetch_probe(int mads, hipStream_t s, u32* out, u32 trips) {
    const dim3 grid(1024), block(64);  // one wave per SIMD, like the lane kernels
    switch (mads) {
        case 1024: hipLaunchKernelGGL(k_ifetch_probe<1024>, grid, block, 0, s, out, trips); break;
        case 8192: hipLaunchKernelGGL(k_ifetch_probe<8192>, grid, block, 0, s, out, trips); break;
        case 32768: hipLaunchKernelGGL(k_ifetch_probe<32768>, grid, block, 0, s, out, trips); break;
        default: hipLaunchKernelGGL(k_ifetch_probe<131072>, grid, block, 0, s, out, trips); break;
    }
}

}  // namespace ecg
