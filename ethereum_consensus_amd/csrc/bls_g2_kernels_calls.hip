// k_sig / k_h2c once more, on the COMPACT-CODE tower (see bls_pairing_kernels_calls.hip).
#define ECG_TOWER_CALLS 1
#define ECG_KN(name) name##_calls
#include "bls_g2_kernels.hip"
