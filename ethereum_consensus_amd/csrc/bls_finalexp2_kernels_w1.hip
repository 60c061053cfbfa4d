// k_finalexp2 once more with the whole register file (ONE wave per SIMD): the build for batches of up to half a round of lanes
// (ECGPU_SPLIT_MAX = 32 768 tuples) and ragged tails of that size, like bls_pairing2_kernels_w1.hip for the Miller loop.
#define ECG_F2_WAVES 1
#define ECG_F2_NAME k_finalexp2_w1
#include "bls_finalexp2_kernels.hip"
