// k_pk_validate once more, with room for 2 waves per SIMD (see bls_g1_kernels.hip).
#define ECG_G1_WAVES 2
#include "bls_g1_kernels.hip"
