// The pairing-check kernels of the BLS batch pipeline (bls.hip launches them): everything above the Fp12 tower lives
// here so that it compiles beside the rest of the pipeline instead of in front of it.
//   k_pairing        lane = tuple   2-pair Miller loop + final exponentiation + status algebra
//   k_miller_pairs   lane = pair    aggregate_verify: one Miller loop per lane
//   k_aggv_final     one lane       product of the Miller values, final exponentiation, status
// (the e(pk, H(m)) == e(g1, sig) equation of /root/reference/ethereum-consensus/src/crypto/bls.rs:71,106,126)
#include "bls_kernels.h"


namespace ecg {

// fast_aggregate_verify tuple i: status algebra + pairing equation.
// k_of: number of keys of tuple i = pk_off ? pk_off[i+1]-pk_off[i] : 1.
__global__ void __launch_bounds__(BLS_BLOCK, ECG_BLS_WAVES) k_pairing(const A1* agg, const u8* st_pk, const u32* pk_off, const A2* hpts,
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

// Second half of the two-lanes-per-tuple pairing check (bls_pairing2_kernels.hip k_miller2 wrote the Miller value of every
// tuple whose status is still 0xff): final exponentiation on one lane per tuple, status.
#if defined(ECG_EXPERIMENTS)  // the one-lane final exponentiation behind k_miller2: default at no size since round 4 (the lane pair runs it)
__global__ void __launch_bounds__(BLS_BLOCK, ECG_BLS_WAVES) k_finalexp(const Fp12* fs, u32 n, u8* status_out) {
    u32 i = blockIdx.x * BLS_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (status_out[i] != 0xff) return;
    Fp12 f = fs[i], e;
    final_exponentiation(e, f);
    status_out[i] = fp12_is_one(e) ? ECGPU_SUCCESS : ECGPU_VERIFY_FAIL;
}
#endif

// ---- aggregate_verify: one Miller loop per lane, product + final exponentiation on one lane ------
__global__ void __launch_bounds__(BLS_BLOCK, ECG_BLS_WAVES) k_miller_pairs(const A1* pts, const A2* hpts, const A2* sigpt, u32 n, Fp12* fs) {
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

// One workgroup of 64 lanes: the first failing key in list order by atomicMin (as k_sum does), the Miller values multiplied in
// 64 stripes (lane t: fs[t] fs[t + 64] ...; in place), then lane 0 multiplies the stripe products and runs the final
// exponentiation: n / 64 + 64 dependent Fp12 products instead of n (round 2: everything on one lane).
__global__ void __launch_bounds__(BLS_BLOCK, ECG_BLS_WAVES) k_aggv_final(const u8* st_pk, u32 n_pks, u32 n_msgs, const u8* st_dec, const u8* st_grp,
                                                           Fp12* fs, u8* status_out) {
    if (blockIdx.x != 0) return;
    __shared__ u32 first_bad;
    __shared__ u32 early;  // 0x100 | status: decided without the pairing product
    const u32 t = threadIdx.x;
    if (t == 0) {
        first_bad = 0xffffffffu;
        early = 0;
    }
    __syncthreads();
    u32 mine = 0xffffffffu;
    for (u32 i = t; i < n_pks && mine == 0xffffffffu; i += BLS_BLOCK)
        if (st_pk[i]) mine = i;
    if (mine != 0xffffffffu) atomicMin(&first_bad, mine);
    __syncthreads();
    if (t == 0) {
        if (first_bad != 0xffffffffu) early = 0x100 | st_pk[first_bad];
        else if (st_dec[0]) early = 0x100 | st_dec[0];
        else if (n_pks == 0 || n_pks != n_msgs) early = 0x100 | ECGPU_VERIFY_FAIL;
        else if (st_grp[0]) early = 0x100 | ECGPU_IN_VERIFY | st_grp[0];  // verify's own group check: Error::InvalidSignature (crypto/bls.rs:106-111)
    }
    __syncthreads();
    if (early) {
        if (t == 0) *status_out = (u8)early;
        return;
    }
    // fs[0 .. n_pks]: n_pks key pairs and the (-g1, sig) pair
    if (t <= n_pks) {
        Fp12 f = fs[t];
        for (u32 i = t + BLS_BLOCK; i <= n_pks; i += BLS_BLOCK) {
            Fp12 g = fs[i];
            fp12_mul(f, f, g);
        }
        fs[t] = f;
    }
    __syncthreads();
    if (t != 0) return;
    Fp12 f = fs[0];
    for (u32 i = 1; i <= n_pks && i < BLS_BLOCK; i++) {
        Fp12 g = fs[i];
        fp12_mul(f, f, g);
    }
    Fp12 e;
    final_exponentiation(e, f);
    *status_out = fp12_is_one(e) ? ECGPU_SUCCESS : ECGPU_VERIFY_FAIL;
}

}  // namespace ecg
