// k_sig / k_h2c once more, on the COMPACT-CODE tower (see bls_pairing_kernels_calls.hip).
#define ECG_TOWER_CALLS 1
#define ECG_LINEAR_CALLS 1  // modular additions as calls as well: the Miller iteration then (nearly) fits the instruction cache
#define ECG_KN(name) name##_calls
#include "bls_g2_kernels.hip"
