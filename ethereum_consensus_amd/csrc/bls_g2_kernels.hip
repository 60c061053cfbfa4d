// The G2 stage kernels of the BLS batch pipeline (bls.hip launches them):
//   k_sig   lane = signature   96 B -> affine G2 (Fp2 sqrt) + psi subgroup check
//   k_h2c   lane = message     hash_to_curve G2 (SHA-256 xmd, SSWU, 3-isogeny, cofactor clearing)
// (Signature::try_from / verify's group check and the hash-to-curve of /root/reference/ethereum-consensus/src/crypto/bls.rs:
// 69-71,330-336.)  bls_g2_kernels_calls.hip compiles this file a second time on the compact-code tower with the kernel
// names suffixed; bls.hip picks the set once per process from the box self-check.
#include "bls_kernels.h"

#ifndef ECG_KN
#define ECG_KN(name) name
#endif

namespace ecg {

__global__ void __launch_bounds__(BLS_BLOCK, ECG_BLS_WAVES) ECG_KN(k_sig)(const u8* sigs96, u32 n, A2* pts, u8* st_dec, u8* st_grp) {
    u32 i = blockIdx.x * BLS_BLOCK + threadIdx.x;
    if (i >= n) return;
    A2 p;
    u8 sd, sg;
    stage_sig(p, sd, sg, sigs96 + 96 * (size_t)i);
    pts[i] = p;
    st_dec[i] = sd;
    st_grp[i] = sg;
}

#if defined(ECG_EXPERIMENTS)  // (round 6) the first form of the row stages -- decoding on one lane, the subgroup check on rows -- is the
                               // default at no size since k_sig_row: experiments library only
// the decoding alone (small batches: the subgroup check then runs on rows, bls_row_g2.hip k_sig_group_row)
__global__ void __launch_bounds__(BLS_BLOCK, ECG_BLS_WAVES) ECG_KN(k_sig_decode)(const u8* sigs96, u32 n, A2* pts, u8* st_dec) {
    u32 i = blockIdx.x * BLS_BLOCK + threadIdx.x;
    if (i >= n) return;
    A2 p;
    const u8 sd = (u8)g2_decompress(p, sigs96 + 96 * (size_t)i);
    pts[i] = p;
    st_dec[i] = sd;
}
#endif

// msg_off == nullptr: message i = msgs + 32 i (32 bytes)
__global__ void __launch_bounds__(BLS_BLOCK, ECG_BLS_WAVES) ECG_KN(k_h2c)(const u8* msgs, const u64* msg_off, u32 n, A2* hpts) {
    u32 i = blockIdx.x * BLS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const u8* m = msg_off ? msgs + msg_off[i] : msgs + 32 * (size_t)i;
    size_t len = msg_off ? (size_t)(msg_off[i + 1] - msg_off[i]) : 32;
    A2 h;
    hash_to_g2(h, m, len);
    hpts[i] = h;
}

// The message stage on two lanes per message: lane 2i + j runs map j of message i (hash_to_g2_map), then one lane per message
// finishes (sum of the two points, cofactor clearing, affine).  For batches that leave SIMDs idle anyway.
__global__ void __launch_bounds__(BLS_BLOCK, ECG_BLS_WAVES) ECG_KN(k_h2c_map)(const u8* msgs, const u64* msg_off, u32 n, J2* maps) {
    u32 t = blockIdx.x * BLS_BLOCK + threadIdx.x;
    if (t >= 2 * n) return;
    const u32 i = t >> 1;
    const u8* m = msg_off ? msgs + msg_off[i] : msgs + 32 * (size_t)i;
    size_t len = msg_off ? (size_t)(msg_off[i + 1] - msg_off[i]) : 32;
    J2 q;
    hash_to_g2_map(q, m, len, (int)(t & 1));
    maps[t] = q;
}
#if defined(ECG_EXPERIMENTS)  // (round 6) the one-lane end of the two-lane message stage (round 3): the lane pair took over in round 4, rows in
                               // round 5 -- the default at no size on any box: experiments library only
__global__ void __launch_bounds__(BLS_BLOCK, ECG_BLS_WAVES) ECG_KN(k_h2c_finish)(const J2* maps, u32 n, A2* hpts) {
    u32 i = blockIdx.x * BLS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const J2 q0 = maps[2 * i], q1 = maps[2 * i + 1];
    A2 h;
    hash_to_g2_finish(h, q0, q1);
    hpts[i] = h;
}
#endif

}  // namespace ecg
