// k_sig / k_h2c once more with room for TWO waves per SIMD (256 registers): for batches of 131 072 tuples and more, where more than
// one wave per SIMD is waiting -- two half-file waves issue more than one full-file wave (a lone wave issues once per ~5 cycles
// whatever it runs).  Measured at 2^20 tuples: k_sig 35.9 -> 29.8 ms, k_h2c 87.2 -> 72.7 ms (profiles/r04o_*).  At 65 536
// tuples there is exactly one wave per SIMD and the full-file build is the faster one.  (The register budget of the callees
// follows the kernel's launch bounds only when the kernel is alone in its translation unit: hence a unit of its own, like
// bls_g1_kernels_w2.hip.)
#define ECG_BLS_WAVES 2
#define ECG_KN(name) name##_w2
#include "bls_g2_kernels.hip"
