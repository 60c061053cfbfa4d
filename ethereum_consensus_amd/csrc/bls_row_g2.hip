// Side stages of a SMALL batch on rows (csrc/bls_rowcurve.h): one point per 16-lane row, four rows to a wave.
//   k_h2c_finish_row   row t = message t: the two mapped points added, cofactor cleared, affine H(m) out
#include "bls_kernels.h"
#include "bls_rowcurve.h"

namespace ecg {

__global__ void __launch_bounds__(64) k_h2c_finish_row(const J2* maps, u32 n, A2* hpts) {
    __shared__ __attribute__((aligned(16))) u32 tab[4][16 * ROW_REG_DW];
    const u32 row = threadIdx.x >> 4, i = blockIdx.x * 4 + row;
    if (i >= n) return;
    r_hash_to_g2_finish(&hpts[i], &maps[2 * (size_t)i], &maps[2 * (size_t)i + 1], tab[row]);
}

void launch_h2c_finish_row(hipStream_t s, const J2* maps, u32 n, A2* hpts) {
    hipLaunchKernelGGL(k_h2c_finish_row, dim3((n + 3) / 4), dim3(64), 0, s, maps, n, hpts);
}

}  // namespace ecg
