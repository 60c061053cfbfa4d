// Host build of the lane programs under -fsanitize=undefined,address (tests/test_sanitize.py): the division-step inversion works
// on SIGNED 30-bit limbs with arithmetic shifts and wrap-around low words, the two-lane message stage passes points by reference --
// the places of this round where undefined behaviour or an out-of-bounds access could hide behind a correct-looking result.
#include "bls_verify.h"
#include <cstdio>
#include <cstdlib>
namespace ecg { unsigned long long g_ecg_fp_mul_count = 0, g_ecg_fp_sqr_count = 0, g_ecg_fp_mad_count = 0, g_ecg_column_overflows = 0; }
using namespace ecg;
int main() {
    // random field elements through fp_inv: a * inv(a) == 1, plus hash_to_g2 on a few messages (exercises sqrt chains, inversions)
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (u32)(s >> 16); };
    int bad = 0;
    for (int it = 0; it < 2000; it++) {
        Fp a;
        for (int i = 0; i < FP_N; i++) a.l[i] = rnd() & FP_MASK;
        a.l[FP_N - 1] &= 0x3fff;
        a = fp_mul(a, blsc::R2);
        Fp inv = fp_inv(a);
        Fp one = fp_mul(a, inv);
        if (!fp_eq(one, fp_one()) && !fp_is_zero(a)) bad++;
    }
    Fp z = fp_zero();
    if (!fp_is_zero(fp_inv(z))) bad++;
    for (int m = 0; m < 4; m++) {
        u8 msg[32];
        for (int i = 0; i < 32; i++) msg[i] = (u8)rnd();
        A2 h, h2;
        hash_to_g2(h, msg, 32);
        J2 q0, q1;
        hash_to_g2_map(q0, msg, 32, 0);
        hash_to_g2_map(q1, msg, 32, 1);
        hash_to_g2_finish(h2, q0, q1);
        if (!fp_eq(h.x.c0, h2.x.c0) || !fp_eq(h.y.c1, h2.y.c1)) bad++;
    }
    printf("bad=%d\n", bad);
    return bad;
}
