// The G1 stage kernel of the BLS batch pipeline (bls.hip launches it):
//   k_pk_validate   lane = public key   48 B -> affine G1 (decompress: Fp sqrt; reject infinity; subgroup check)
// (PublicKey::try_from of /root/reference/ethereum-consensus/src/crypto/bls.rs:279-285, once per key of every call.)
// G1 work is Fp-only -- no tower values -- so it fits a half, a third or a quarter of a SIMD's register file, and committee
// batches bring 8+ waves per SIMD of keys.  The register budget of the callees follows the kernel's launch bounds only when the
// kernel is alone in its translation unit, so every occupancy target is a unit of its own: bls_g1_kernels_w2.hip
// compiles this file again with ECG_G1_WAVES = 2; bls.hip picks by batch size.
#include "bls_kernels.h"

#ifndef ECG_G1_WAVES
#define ECG_G1_WAVES 1
#endif
#define ECG_G1_CAT2(a, b) a##b
#define ECG_G1_CAT(a, b) ECG_G1_CAT2(a, b)
#define ECG_G1_KN(name) ECG_G1_CAT(name##_w, ECG_G1_WAVES)
// the register budget the compiler is given (default: room for ECG_G1_WAVES waves per SIMD); tools/build_variant.sh sets it apart
// from the kernel's name to measure other budgets under the same dispatch
#ifndef ECG_G1_OCCUPANCY
#define ECG_G1_OCCUPANCY ECG_G1_WAVES
#endif

namespace ecg {

__global__ void __launch_bounds__(BLS_BLOCK, ECG_G1_OCCUPANCY) ECG_G1_KN(k_pk_validate)(const u8* pks48, u32 n, A1* pts, u8* st) {
    u32 i = blockIdx.x * BLS_BLOCK + threadIdx.x;
    if (i >= n) return;
    A1 p;
    u8 s = stage_pk_validate(p, pks48 + 48 * (size_t)i);
    pts[i] = p;
    st[i] = s;
}

#if ECG_G1_WAVES == 1 && defined(ECG_EXPERIMENTS)  // (round 6: the first form of the row stages, like k_sig_decode)
// the decoding alone (small batches: the subgroup check then runs on rows, bls_row_g2.hip k_pk_group_row): the status
// key_validate would return before its group check
__global__ void __launch_bounds__(BLS_BLOCK, ECG_G1_OCCUPANCY) k_pk_decode_w1(const u8* pks48, u32 n, A1* pts, u8* st) {
    u32 i = blockIdx.x * BLS_BLOCK + threadIdx.x;
    if (i >= n) return;
    A1 p;
    int s = g1_decompress(p, pks48 + 48 * (size_t)i);
    if (!s && p.inf) s = ECGPU_PK_IS_INFINITY;
    pts[i] = p;
    st[i] = (u8)s;
}
#endif

}  // namespace ecg
