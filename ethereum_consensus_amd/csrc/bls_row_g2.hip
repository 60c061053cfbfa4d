// Side stages of a SMALL batch on rows (csrc/bls_rowcurve.h): one point per 16-lane row, four rows to a wave.
//   k_h2c_finish_row   row t = message t: the two mapped points added, cofactor cleared, affine H(m) out
#include "bls_kernels.h"
#include "bls_rowcurve.h"

namespace ecg {

// a ROW PAIR per message (bls_rowpair.h: one component of every Fp2 coordinate per row): 126 doublings of ~1 500 instructions
__global__ void __launch_bounds__(64) k_h2c_finish_row(const J2* maps, u32 n, A2* hpts) {
    __shared__ __attribute__((aligned(16))) u32 tab[4][16 * ROW_REG_DW];
    const u32 row = threadIdx.x >> 4, i = blockIdx.x * 2 + (row >> 1);
    if (i >= n) return;
    r_hash_to_g2_finish<RP2>(&hpts[i], &maps[2 * (size_t)i], &maps[2 * (size_t)i + 1], tab[row]);
}

// ONE message per wave: both row pairs hold the point, each of the 126 doublings runs its six products three deep over the two
// pairs (bls_rowcurve.h jac_dbl_quad) -- for the few messages of a lone call or a block, whose second pair would sit idle
__global__ void __launch_bounds__(64) k_h2c_finish_quad(const J2* maps, u32 n, A2* hpts) {
    __shared__ __attribute__((aligned(16))) u32 tab[4][16 * ROW_REG_DW];
    const u32 row = threadIdx.x >> 4, i = blockIdx.x;
    if (i >= n) return;
    r_hash_to_g2_finish_quad(&hpts[i], &maps[2 * (size_t)i], &maps[2 * (size_t)i + 1], tab[row]);
}

// row t = map (t & 1) of message t >> 1: expand_message_xmd + field element + 1 / tv2 by the one-lane routines in every lane, the
// SSWU map and the 3-isogeny on the row (two exponentiations of 0.14 ms instead of 0.46)
__global__ void __launch_bounds__(64) k_h2c_map_row(const u8* msgs, const u64* msg_off, u32 n, J2* maps) {
    __shared__ __attribute__((aligned(16))) u32 tab[4][16 * ROW_REG_DW];
    const u32 row = threadIdx.x >> 4, t = blockIdx.x * 4 + row;
    if (t >= 2 * n) return;
    const u32 i = t >> 1;
    const u8* m = msg_off ? msgs + msg_off[i] : msgs + 32 * (size_t)i;
    const size_t len = msg_off ? (size_t)(msg_off[i + 1] - msg_off[i]) : 32;
    r_hash_to_g2_map(&maps[t], m, len, (int)(t & 1), tab[row]);
}
#if defined(ECG_EXPERIMENTS)  // (round 6) the first form of the row stages: subgroup checks of points decoded on one lane each
// row t = signature t, already decoded (k_sig_decode): the psi subgroup check of verify (crypto/bls.rs:71,126)
__global__ void __launch_bounds__(64) k_sig_group_row(const A2* pts, const u8* st_dec, u32 n, u8* st_grp) {
    const u32 row = threadIdx.x >> 4, i = blockIdx.x * 2 + (row >> 1);  // a row pair per signature
    if (i >= n) return;
    u8 g = 0;
    if (st_dec[i] == 0 && !r_g2_in_subgroup<RP2>(&pts[i])) g = ECGPU_POINT_NOT_IN_GROUP;
    if ((threadIdx.x & 31u) == 0) st_grp[i] = g;
}

// row t = public key t, already decoded (k_pk_decode): reject infinity, then the endomorphism subgroup check of key_validate
// (crypto/bls.rs:279-285)
__global__ void __launch_bounds__(64) k_pk_group_row(const A1* pts, u32 n, u8* st) {
    const u32 row = threadIdx.x >> 4, i = blockIdx.x * 4 + row;
    if (i >= n) return;
    if (st[i] != 0) return;  // (the decoder's verdict stands: bad encoding, not on the curve, x == 0, infinity)
    if (!r_g1_in_subgroup(&pts[i]) && (threadIdx.x & 15u) == 0) st[i] = ECGPU_POINT_NOT_IN_GROUP;
}
void launch_pk_group_row(hipStream_t s, const A1* pts, u32 n, u8* st) {
    hipLaunchKernelGGL(k_pk_group_row, dim3((n + 3) / 4), dim3(64), 0, s, pts, n, st);
}
void launch_sig_group_row(hipStream_t s, const A2* pts, const u8* st_dec, u32 n, u8* st_grp) {
    hipLaunchKernelGGL(k_sig_group_row, dim3((n + 1) / 2), dim3(64), 0, s, pts, st_dec, n, st_grp);
}
#endif
// (round 5, last) decoding AND group check of a signature on a row pair, of a key on a row: the square roots -- two Fp
// exponentiations for a signature, one for a key: 0.97 / 0.45 ms on one lane -- run on the row as the SSWU map's do
__global__ void __launch_bounds__(64) k_sig_row(const u8* sigs96, u32 n, A2* pts, u8* st_dec, u8* st_grp) {
    __shared__ __attribute__((aligned(16))) u32 tab[4][16 * ROW_REG_DW];
    const u32 row = threadIdx.x >> 4, i = blockIdx.x * 2 + (row >> 1);
    if (i >= n) return;
    r_sig_decode_and_group(&pts[i], &st_dec[i], &st_grp[i], sigs96 + 96 * (size_t)i, tab[row]);
}
__global__ void __launch_bounds__(64) k_pk_row(const u8* pks48, u32 n, A1* pts, u8* st) {
    __shared__ __attribute__((aligned(16))) u32 tab[4][16 * ROW_REG_DW];
    const u32 row = threadIdx.x >> 4, i = blockIdx.x * 4 + row;
    if (i >= n) return;
    r_pk_validate(&pts[i], &st[i], pks48 + 48 * (size_t)i, tab[row]);
}
void launch_sig_row(hipStream_t s, const u8* sigs96, u32 n, A2* pts, u8* st_dec, u8* st_grp) {
    hipLaunchKernelGGL(k_sig_row, dim3((n + 1) / 2), dim3(64), 0, s, sigs96, n, pts, st_dec, st_grp);
}
void launch_pk_row(hipStream_t s, const u8* pks48, u32 n, A1* pts, u8* st) {
    hipLaunchKernelGGL(k_pk_row, dim3((n + 3) / 4), dim3(64), 0, s, pks48, n, pts, st);
}
void launch_h2c_map_row(hipStream_t s, const u8* msgs, const u64* msg_off, u32 n, J2* maps) {
    hipLaunchKernelGGL(k_h2c_map_row, dim3((2 * n + 3) / 4), dim3(64), 0, s, msgs, msg_off, n, maps);
}
void launch_h2c_finish_quad(hipStream_t s, const J2* maps, u32 n, A2* hpts) {
    hipLaunchKernelGGL(k_h2c_finish_quad, dim3(n), dim3(64), 0, s, maps, n, hpts);
}
void launch_h2c_finish_row(hipStream_t s, const J2* maps, u32 n, A2* hpts) {
    hipLaunchKernelGGL(k_h2c_finish_row, dim3((n + 1) / 2), dim3(64), 0, s, maps, n, hpts);
}

}  // namespace ecg
