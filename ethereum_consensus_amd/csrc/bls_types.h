// BLS12-381 value types shared by the gfx950 kernels and (for the CPU test-suite only) the host
// lane simulator.
//
// A base-field element is 13 x 30-bit little-endian limbs (one u32 each) in Montgomery form with
// R = 2^390, kept "almost reduced": limbs < 2^30, value < 2p.  Why 30-bit limbs on gfx950: the only
// full multiplier is v_mad_u64_u32 (32x32 + 64 -> 64; measured 31 T lane-ops/s,
// profiles/r01a_int_issue_rate_microbench.txt) and it has no usable carry chain from C++.  With
// unsaturated limbs a 64-bit accumulator absorbs 15 products (15 * 2^60 < 2^64) with no carry
// handling, so a Montgomery product is 13 x 26 back-to-back independent multiply-adds into
// accumulator pairs that never move (measured: the saturated 12 x 32-bit CIOS compiled to 288 mads
// + 619 v_mov + 297 64-bit adds; this form to 351 mads + ~200 cheap ops).  One Fp is 13 VGPRs.
#pragma once
#include "common.h"

#if defined(__HIPCC__)
#define ECG_CONST static __device__ constexpr
#else
#define ECG_CONST static constexpr
#endif

namespace ecg {

struct Fp {
    u32 l[13];
};
struct Fp2 {  // c0 + c1 i, i^2 = -1
    Fp c0, c1;
};
struct Fp6 {  // c0 + c1 v + c2 v^2, v^3 = xi = 1 + i
    Fp2 c0, c1, c2;
};
struct Fp12 {  // c0 + c1 w, w^2 = v
    Fp6 c0, c1;
};

}  // namespace ecg
