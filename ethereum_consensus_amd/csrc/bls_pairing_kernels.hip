// The pairing-check kernels of the BLS batch pipeline (bls.hip launches them): everything above the Fp12 tower lives
// here so that it compiles beside the rest of the pipeline instead of in front of it.
//   k_pairing        lane = tuple   2-pair Miller loop + final exponentiation + status algebra
//   k_miller_pairs   lane = pair    aggregate_verify: one Miller loop per lane
//   k_aggv_final     one lane       product of the Miller values, final exponentiation, status
// (the e(pk, H(m)) == e(g1, sig) equation of /root/reference/ethereum-consensus/src/crypto/bls.rs:71,106,126)
#include "bls_kernels.h"

// bls_pairing_kernels_calls.hip compiles this file a second time with -DECG_TOWER_CALLS semantics (the compact-code tower)
// and the kernel names suffixed.
#ifndef ECG_KN
#define ECG_KN(name) name
#endif

namespace ecg {

// fast_aggregate_verify tuple i: status algebra + pairing equation.
// k_of: number of keys of tuple i = pk_off ? pk_off[i+1]-pk_off[i] : 1.
__global__ void __launch_bounds__(BLS_BLOCK, ECG_BLS_WAVES) ECG_KN(k_pairing)(const A1* agg, const u8* st_pk, const u32* pk_off, const A2* hpts,
                                                        const A2* sigpts, const u8* st_dec, const u8* st_grp, const u8* sigs96,
                                                        u32 n, int eth_variant, u8* status_out) {
    u32 i = blockIdx.x * BLS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const u32 k = pk_off ? pk_off[i + 1] - pk_off[i] : 1;
    const bool sig_inf_bytes = sig_is_infinity_bytes(sigs96 + 96 * (size_t)i);
    const bool agg_inf = agg[i].inf != 0;
    u8 pre = combine_fav_status(k, eth_variant != 0, sig_inf_bytes, st_pk[i], st_dec[i], st_grp[i], agg_inf, 0xff);
    if (pre != 0xff) {
        status_out[i] = pre;
        return;
    }
    A1 a = agg[i];
    A2 h = hpts[i];
    A2 s = sigpts[i];
    status_out[i] = stage_pairing(a, h, s);
}

// ---- aggregate_verify: one Miller loop per lane, product + final exponentiation on one lane ------
__global__ void __launch_bounds__(BLS_BLOCK, ECG_BLS_WAVES) ECG_KN(k_miller_pairs)(const A1* pts, const A2* hpts, const A2* sigpt, u32 n, Fp12* fs) {
    u32 i = blockIdx.x * BLS_BLOCK + threadIdx.x;
    if (i > n) return;
    MillerPair pr;
    if (i < n) {
        A1 p = pts[i];
        A2 q = hpts[i];
        miller_pair_init(pr, p, q);
    } else {
        A1 ng;
        ng.x = blsc::G1_X;
        ng.y = blsc::G1_NEG_Y;
        ng.inf = 0;
        A2 s = *sigpt;
        miller_pair_init(pr, ng, s);
    }
    Fp12 f;
    miller_loop(f, &pr, 1);
    fs[i] = f;
}

__global__ void __launch_bounds__(BLS_BLOCK, ECG_BLS_WAVES) ECG_KN(k_aggv_final)(const u8* st_pk, u32 n_pks, u32 n_msgs, const u8* st_dec, const u8* st_grp,
                                                           const Fp12* fs, u8* status_out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (u32 i = 0; i < n_pks; i++)
        if (st_pk[i]) {
            *status_out = st_pk[i];
            return;
        }
    if (st_dec[0]) {
        *status_out = st_dec[0];
        return;
    }
    if (n_pks == 0 || n_pks != n_msgs) {
        *status_out = ECGPU_VERIFY_FAIL;
        return;
    }
    if (st_grp[0]) {
        *status_out = ECGPU_IN_VERIFY | st_grp[0];  // verify's own group check: Error::InvalidSignature (crypto/bls.rs:106-111)
        return;
    }
    Fp12 f = fs[0];
    for (u32 i = 1; i <= n_pks; i++) {
        Fp12 g = fs[i];
        fp12_mul(f, f, g);
    }
    Fp12 e;
    final_exponentiation(e, f);
    *status_out = fp12_is_one(e) ? ECGPU_SUCCESS : ECGPU_VERIFY_FAIL;
}

}  // namespace ecg
