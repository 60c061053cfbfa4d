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

// Out-of-line lane routines take their operands by reference, i.e. through GENERIC pointers: the compiler must use flat
// loads / stores, and because flat accesses may return out of order between the LDS and the memory path every use waits for
// ALL outstanding accesses (s_waitcnt 0) -- at one wave per SIMD that is a full 1-2 us private-segment round trip per
// access group (measured: 90 k of the 244 k cycles of an out-of-line Fp12 product).  Every such object in this code base is
// a kernel- or function-local, i.e. lives in the private segment; these helpers say so: the copy goes through
// address-space-5 pointers (scratch_load / scratch_store with their own counters, issued early, waited for one by one).
// ONLY for references to locals -- never for global or LDS memory.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
ECG_D T ecg_priv_load(const T& x) {
    static_assert(sizeof(T) % 4 == 0 && alignof(T) >= 4, "dword-granular objects only");
    typedef __attribute__((address_space(5))) const u32* src_t;
    T r;
    src_t p = (src_t)(const u32*)(const void*)&x;
    u32* q = (u32*)&r;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) q[i] = p[i];
    // (materialising every dword here with an asm barrier, so that the wave parks once per object instead of once per first
    // use, was measured and does not pay: 44.5 against 43.8 ms per 65 536 tuples, profiles/r02u_staggered_start.txt)
    // (a scheduling barrier after the loads of small objects, which keeps them together ahead of their uses, halves the static
    // wait count of an out-of-line doubling and changes nothing measurable: 41.6 ms either way)
    return r;
}
template <class T>
ECG_D void ecg_priv_store(T& x, const T& v) {
    static_assert(sizeof(T) % 4 == 0 && alignof(T) >= 4, "dword-granular objects only");
    typedef __attribute__((address_space(5))) u32* dst_t;
    dst_t p = (dst_t)(u32*)(void*)&x;
    const u32* q = (const u32*)&v;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) p[i] = q[i];
}
// COMPILER BUG WORKAROUND (ROCm 7.2 / LLVM AMDGPU branch relaxation).  A branch whose distance exceeds the 16-bit offset
// (128 KB of code: our loops over inlined tower arithmetic) is expanded to s_getpc_b64 / s_add_u32 / s_addc_u32 /
// s_setpc_b64 through a scavenged SGPR pair, and in a function that makes NO call the scavenger hands out s[30:31] -- the
// function's own return address: the return then jumps back into the loop and the kernel never terminates (round 2 met this
// as "a kernel that does not terminate on the device" and backed off; tools/isa_census.py --check-long-branches finds it
// statically).  A function that contains a call saves s[30:31] in its prologue and restores it before returning, so a clobber
// in between is harmless: every out-of-line lane routine with a loop larger than the branch range calls this once.
static __device__ __attribute__((noinline)) void ecg_force_return_address_save() { asm volatile("" ::: "memory"); }
#define ECG_LONG_BRANCH_GUARD() ecg_force_return_address_save()
#else
#define ECG_LONG_BRANCH_GUARD() ((void)0)
template <class T>
ECG_HD T ecg_priv_load(const T& x) { return x; }
template <class T>
ECG_HD void ecg_priv_store(T& x, const T& v) { x = v; }
#endif
