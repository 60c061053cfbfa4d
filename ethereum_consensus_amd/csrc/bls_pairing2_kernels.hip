// The Miller loop of the K = 1 / fast_aggregate_verify pairing check on TWO lanes per tuple (bls_pair2.h): a translation unit
// of its own because the kernel's launch bounds (two waves per SIMD: 256 registers) govern the register budget of everything
// it calls, and because it owns 6 LDS lane slots per lane where the one-lane kernels of bls_pairing_kernels.hip own 12.
//   k_miller2    lanes 2t, 2t + 1 = tuple t: status algebra, 2-pair Miller loop -> the Miller value f in memory
//   (k_finalexp, bls_pairing_kernels.hip: one lane per tuple, final exponentiation of f, status)
// (the e(pk, H(m)) == e(g1, sig) equation of /root/reference/ethereum-consensus/src/crypto/bls.rs:71,126)
// bls_pairing2_kernels_w1.hip compiles this file again with the whole register file (k_miller2_w1): up to half a round of lanes
// the two lanes per tuple are still ONE wave per SIMD, and 512 registers hold what 256 spill (17.6 GB of private-segment traffic
// per 65 536-tuple launch, profiles/r04v3split_*).
#define ECG_LANE_SLOTS 6
#ifndef ECG_M2_WAVES
#define ECG_M2_WAVES 2
#define ECG_M2_NAME k_miller2
#endif
#define ECG_BLS_WAVES ECG_M2_WAVES
// (round 6) the two-wave build is the default at no batch size on any box (17.6 GB of private-segment traffic per
// 65 536-tuple launch): it is compiled only into the experiments library (ECGPU_EXPERIMENTS=1 python -m ethereum_consensus_amd.build)
#if ECG_M2_WAVES == 1 || defined(ECG_EXPERIMENTS)
#include "bls_kernels.h"
#include "bls_pair2.h"

namespace ecg {

__global__ void __launch_bounds__(BLS_BLOCK, ECG_M2_WAVES) ECG_M2_NAME(const A1* agg, const u8* st_pk, const u32* pk_off, const A2* hpts, const A2* sigpts,
                                                         const u8* st_dec, const u8* st_grp, const u8* sigs96, u32 n, int eth_variant,
                                                         u8* status_out, Fp12* fs) {
    const u32 lane = blockIdx.x * BLS_BLOCK + threadIdx.x;
    const u32 i = lane >> 1;
    if (i >= n) return;
    const u32 k = pk_off ? pk_off[i + 1] - pk_off[i] : 1;
    const bool sig_inf_bytes = sig_is_infinity_bytes(sigs96 + 96 * (size_t)i);
    const bool agg_inf = agg[i].inf != 0;
    const u8 pre = combine_fav_status(k, eth_variant != 0, sig_inf_bytes, st_pk[i], st_dec[i], st_grp[i], agg_inf, 0xff);
    if ((lane & 1) == 0) status_out[i] = pre;  // 0xff: the pairing equation decides (k_finalexp)
    if (pre != 0xff) return;                   // both lanes of the pair leave together
    MillerPairH pr[2];
    {
        const A1 a = agg[i];
        const A2 h = hpts[i];
        miller_pair_h_init(pr[0], a, h);
    }
    {
        A1 ng;
        ng.x = blsc::G1_X;
        ng.y = blsc::G1_NEG_Y;
        ng.inf = 0;
        const A2 s = sigpts[i];
        miller_pair_h_init(pr[1], ng, s);
    }
    H12 f;
    h_miller_loop(f, pr);
    h12_store(&fs[i], f);
}

}  // namespace ecg
#endif
