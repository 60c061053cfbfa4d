// The final exponentiation of the K = 1 / fast_aggregate_verify pairing check on TWO lanes per tuple (bls_finalexp2.h), behind
// k_miller2 (bls_pairing2_kernels.hip): a translation unit of its own because the kernel's launch bounds govern the register budget
// of everything it calls, and because it owns 6 LDS lane slots per lane where the one-lane kernels own 12.
//   k_finalexp2     two waves per SIMD (256 registers): any batch size
//   k_finalexp2_w1  (bls_finalexp2_kernels_w1.hip compiles this file again) the whole register file: up to half a round of lanes,
//                   where the two lanes per tuple are still ONE wave per SIMD
// (the e(pk, H(m)) == e(g1, sig) equation of /root/reference/ethereum-consensus/src/crypto/bls.rs:71,126)
#define ECG_LANE_SLOTS 6
#ifndef ECG_F2_WAVES
#define ECG_F2_WAVES 2
#define ECG_F2_NAME k_finalexp2
#endif
#define ECG_BLS_WAVES ECG_F2_WAVES
// (round 6) the two-wave build runs at no batch size by default (11.6 ms against the one-lane kernel's 10.4, profiles/r04f2_*): compiled
// only into the experiments library, like the two-wave k_miller2
#if ECG_F2_WAVES == 1 || defined(ECG_EXPERIMENTS)
#include "bls_kernels.h"
#include "bls_finalexp2.h"

namespace ecg {

__global__ void __launch_bounds__(BLS_BLOCK, ECG_F2_WAVES) ECG_F2_NAME(const Fp12* fs, u32 n, u8* status_out) {
    const u32 lane = blockIdx.x * BLS_BLOCK + threadIdx.x;
    const u32 i = lane >> 1;
    if (i >= n) return;
    if (status_out[i] != 0xff) return;  // both lanes of the pair leave together (the status is the Miller kernel's)
    H12 f, e;
    h12_load(f, &fs[i]);
    h_final_exponentiation(e, f);
    const bool one = h12_is_one(e);
    // (the two lanes of a pair sit in one wave and both read the status byte above before either reaches this write)
    if ((lane & 1) == 0) status_out[i] = one ? ECGPU_SUCCESS : ECGPU_VERIFY_FAIL;
}

}  // namespace ecg
#endif
