// The end of the message stage on TWO lanes per message (bls_g2_pair2.h) for the latency-bound small batches:
//   k_h2c_finish2    lanes 2t, 2t + 1 = message t: the two mapped points added, cofactor cleared, affine H(m) out
// (the hash_to_curve step blst performs inside every verify call: /root/reference/ethereum-consensus/src/crypto/bls.rs:71,126)
#include "bls_kernels.h"
#include "bls_g2_pair2.h"

namespace ecg {

__global__ void __launch_bounds__(BLS_BLOCK, ECG_BLS_WAVES) k_h2c_finish2(const J2* maps, u32 n, A2* hpts) {
    const u32 lane = blockIdx.x * BLS_BLOCK + threadIdx.x;
    const u32 i = lane >> 1;
    if (i >= n) return;
    h_hash_to_g2_finish(&hpts[i], maps[2 * (size_t)i], maps[2 * (size_t)i + 1]);
}

}  // namespace ecg
