// Common definitions for the gfx950 kernels.
//
// Every arithmetic routine under csrc/ is written as ECG_HD (`__host__ __device__`) so that the
// *same source the GPU runs* can also be compiled by g++ into tests/hostsim/ -- a kernel
// simulator used ONLY by the CPU test-suite to check the device arithmetic against oracle/
// without a GPU.  The product library (libecgpu.so) contains no host execution path: every
// ecgpu_* entry point launches HIP kernels and fails with ECGPU_ERR_NO_DEVICE otherwise.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ECG_HD __host__ __device__ __forceinline__
#define ECG_HD_NOINLINE static __host__ __device__ __attribute__((noinline))
#define ECG_D __device__ __forceinline__
#else
#define ECG_HD inline
#define ECG_HD_NOINLINE static __attribute__((noinline))
#define ECG_D inline
#endif

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;

ECG_HD u32 ecg_bswap32(u32 x) { return __builtin_bswap32(x); }

// 3-input bitwise ops: gfx950 v_bitop3_b32 (truth table indexed by {S0,S1,S2} with
// S0=0xF0, S1=0xCC, S2=0xAA); plain C on the host simulator.
ECG_HD u32 ecg_xor3(u32 a, u32 b, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}
ECG_HD u32 ecg_sel(u32 m, u32 a, u32 b) {  // m ? a : b, bitwise
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA);
#else
    return (m & a) | (~m & b);
#endif
}
ECG_HD u32 ecg_maj(u32 a, u32 b, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);
#else
    return (a & b) | (c & (a | b));
#endif
}
