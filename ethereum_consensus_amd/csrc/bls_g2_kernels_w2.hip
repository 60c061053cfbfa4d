// k_sig / k_h2c once more, with room for two waves per SIMD (half the register file each): a signature stage and a message
// stage side by side on one SIMD (bls.hip, ECGPU_G2_WAVES).
#define ECG_BLS_WAVES 2
#define ECG_KN(name) name##_w2
#include "bls_g2_kernels.hip"
