// Launch geometry of the BLS lane kernels and the kernels that live in a translation unit of their own.
#pragma once
#include "bls_verify.h"

namespace ecg {

constexpr int BLS_BLOCK = 64;  // one wave per workgroup: spreads small batches over every CU
#if defined(ECG_LANE_SLOT_BLOCK)
// the LDS lane slots of bls_pairing.h are indexed by threadIdx.x & 63: a workgroup of the pairing kernels must be one wave
static_assert(BLS_BLOCK == ECG_LANE_SLOT_BLOCK, "the pairing kernels' lane slots assume one wave per workgroup");
#endif
#ifndef ECG_BLS_WAVES
#define ECG_BLS_WAVES 1  // waves per SIMD the register allocator must leave room for: 1 = the whole 512-entry VGPR+AGPR file.
                         // These lane kernels hold hundreds of live field-element limbs, so registers beat occupancy
                         // (re-measured after every restructuring)
#endif


// bls_g1_kernels.hip / bls_g1_kernels_w2.hip: the key stage with room for one / two waves per SIMD
__global__ void k_pk_validate_w1(const u8* pks48, u32 n, A1* pts, u8* st);
__global__ void k_pk_validate_w2(const u8* pks48, u32 n, A1* pts, u8* st);

// bls_g2_kernels.hip / bls_g2_kernels_calls.hip
__global__ void k_sig(const u8* sigs96, u32 n, A2* pts, u8* st_dec, u8* st_grp);
__global__ void k_h2c(const u8* msgs, const u64* msg_off, u32 n, A2* hpts);
__global__ void k_h2c_map(const u8* msgs, const u64* msg_off, u32 n, J2* maps);
__global__ void k_h2c_finish2(const J2* maps, u32 n, A2* hpts);  // bls_g2_pair2_kernels.hip: two lanes per message
__global__ void k_h2c_map_calls(const u8* msgs, const u64* msg_off, u32 n, J2* maps);
// bls_g2_kernels_w2.hip: room for two waves per SIMD (batches beyond 65 536 tuples)
__global__ void k_sig_w2(const u8* sigs96, u32 n, A2* pts, u8* st_dec, u8* st_grp);
__global__ void k_h2c_w2(const u8* msgs, const u64* msg_off, u32 n, A2* hpts);
__global__ void k_h2c_map_w2(const u8* msgs, const u64* msg_off, u32 n, J2* maps);
__global__ void k_h2c_finish_w2(const J2* maps, u32 n, A2* hpts);
__global__ void k_sig_calls(const u8* sigs96, u32 n, A2* pts, u8* st_dec, u8* st_grp);
__global__ void k_h2c_calls(const u8* msgs, const u64* msg_off, u32 n, A2* hpts);

// bls_row_g2.hip: side stages of a small batch with one point per 16-lane row
void launch_h2c_finish_row(hipStream_t s, const J2* maps, u32 n, A2* hpts);
void launch_h2c_finish_quad(hipStream_t s, const J2* maps, u32 n, A2* hpts);  // one message per wave, doublings over both row pairs
void launch_h2c_map_row(hipStream_t s, const u8* msgs, const u64* msg_off, u32 n, J2* maps);
void launch_sig_row(hipStream_t s, const u8* sigs96, u32 n, A2* pts, u8* st_dec, u8* st_grp);  // decoding + group check on a row pair
void launch_pk_row(hipStream_t s, const u8* pks48, u32 n, A1* pts, u8* st);                    // key_validate on a row
#if defined(ECG_EXPERIMENTS)  // first forms that are the default at no size (round 6: experiments library only)
__global__ void k_h2c_finish(const J2* maps, u32 n, A2* hpts);        // the one-lane end of the two-lane message stage (round 3)
__global__ void k_h2c_finish_calls(const J2* maps, u32 n, A2* hpts);
void launch_sig_group_row(hipStream_t s, const A2* pts, const u8* st_dec, u32 n, u8* st_grp);
void launch_pk_group_row(hipStream_t s, const A1* pts, u32 n, u8* st);
__global__ void k_pk_decode_w1(const u8* pks48, u32 n, A1* pts, u8* st);  // bls_g1_kernels.hip: decoding + infinity alone
__global__ void k_sig_decode(const u8* sigs96, u32 n, A2* pts, u8* st_dec);  // bls_g2_kernels.hip: Signature::try_from alone
__global__ void k_sig_decode_calls(const u8* sigs96, u32 n, A2* pts, u8* st_dec);
#endif

// bls_pairing_kernels.hip
__global__ void k_pairing(const A1* agg, const u8* st_pk, const u32* pk_off, const A2* hpts, const A2* sigpts, const u8* st_dec,
                          const u8* st_grp, const u8* sigs96, u32 n, int eth_variant, u8* status_out);
#if defined(ECG_EXPERIMENTS)
__global__ void k_finalexp(const Fp12* fs, u32 n, u8* status_out);
__global__ void k_finalexp2(const Fp12* fs, u32 n, u8* status_out);
#endif
__global__ void k_finalexp2_w1(const Fp12* fs, u32 n, u8* status_out);
// bls_pairing2_kernels.hip: the same check's Miller loop on two lanes per tuple, two waves per SIMD (bls_pair2.h)
#if defined(ECG_EXPERIMENTS)
__global__ void k_miller2(const A1* agg, const u8* st_pk, const u32* pk_off, const A2* hpts, const A2* sigpts, const u8* st_dec,
                          const u8* st_grp, const u8* sigs96, u32 n, int eth_variant, u8* status_out, Fp12* fs);
#endif
__global__ void k_miller2_w1(const A1* agg, const u8* st_pk, const u32* pk_off, const A2* hpts, const A2* sigpts, const u8* st_dec,
                          const u8* st_grp, const u8* sigs96, u32 n, int eth_variant, u8* status_out, Fp12* fs);
__global__ void k_miller_pairs(const A1* pts, const A2* hpts, const A2* sigpt, u32 n, Fp12* fs);
__global__ void k_aggv_final(const u8* st_pk, u32 n_pks, u32 n_msgs, const u8* st_dec, const u8* st_grp, Fp12* fs, u8* status_out);

}  // namespace ecg
