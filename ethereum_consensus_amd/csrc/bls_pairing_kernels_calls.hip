// The pairing-check kernels once more, on the COMPACT-CODE tower (Karatsuba over out-of-line Fp products, ECG_TOWER_CALLS in
// bls_fp.h / bls_tower.h): 15 % slower than the sums-of-products kernels on a healthy box, but their hot loops are a
// fraction of the code, and on a box whose instruction fetch does not keep up beyond the 64 KB instruction cache
//  they are the faster ones.  bls.hip picks the set once per process from the box self-check
// (ECGPU_TOWER=sums|calls overrides).
#define ECG_TOWER_CALLS 1
#define ECG_LINEAR_CALLS 1  // modular additions as calls as well: the Miller iteration then (nearly) fits the instruction cache
#define ECG_KN(name) name##_calls
#include "bls_pairing_kernels.hip"
