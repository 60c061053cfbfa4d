// k_miller2 once more with the whole register file (ONE wave per SIMD): the build for batches of up to half a round of lanes
// (ECGPU_SPLIT_MAX = 32 768 tuples = 65 536 lanes) and for ragged tails of that size, where a second wave per SIMD never arrives
// and the 256-register build only spills.  (A unit of its own: the callees' register budget follows the kernel's launch bounds
// only when the kernel is alone in its translation unit.)
#define ECG_M2_WAVES 1
#define ECG_M2_NAME k_miller2_w1
#include "bls_pairing2_kernels.hip"
