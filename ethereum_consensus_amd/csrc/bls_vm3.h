// The "sum-of-products VM": lane-group execution of straight-line programs over an LDS-resident Fp register file
// (tools/gen_bls_vm3.py has the why, the programs and the encoding).
//
// A tuple (one pairing check) is owned by ECG_VM3_<part>_LANES consecutive lanes of a wave (16 in the Miller loops, 12 in the final exponentiation).  A program is a sequence of rounds;
// round r has one header word (N | nder << 8) and gives lane slot k the 8-dword descriptor prog[(r * LANES + k) * 8 ..]:
//     w0: dst | a0 << 8 | a1 << 16 | a2 << 24     w1: a3 | a4 << 8 | a5 << 16 | a6 << 24
//     w2: b0 | b1 << 8 | b2 << 16 | b3 << 24      w3: b4 | b5 << 8 | b6 << 16
//     w4 + d (d < 4): derived output d: reg | (c_own & 255) << 8 | (c_partner & 255) << 16 | K << 24   (reg 0 = none)
// The lane computes own = sum_{k < N} R[a_k] * R[b_k] (Montgomery, one reduction: fp_sumprod<N>; N = 0: own = R[a0]),
// stores it to dst, exchanges `own` with its pair partner (lane ^ 1: the other component of the same Fp2 value) and stores
// up to nder derived registers c_own * own + c_partner * partner + K p -- limbs renormalised, NOT reduced modulo p: they
// are product operands of later rounds (bounds checked by the generator).  Register 0 is ZERO: unused operand slots read
// it (0 * 0 adds nothing to a sum) and a store to it is skipped (idle lanes, results nobody reads).  Register numbers from
// ECG_VM3_CONST_BASE up name the program's constants: one copy per workgroup behind the tuples' slices, not one per tuple.
// All reads of a round happen before its writes, so a register may be reused by the round that last reads it.
#pragma once
#include "bls_fp.h"

namespace ecg {

constexpr u32 VM3_REG_DW = 13;  // dwords per Fp register
constexpr u32 VM3_DESC_DW = 8;  // dwords per lane descriptor
constexpr u32 VM3_CONST_BASE = 192;

// the register file of one tuple: its own slice and the workgroup's constants
struct Vm3Regs {
    u32* own;
    const u32* consts;
};

ECG_HD Fp vm3_load(const Vm3Regs& R, u32 r) {
    Fp x;
    const u32* p = r >= VM3_CONST_BASE ? R.consts + (r - VM3_CONST_BASE) * VM3_REG_DW : R.own + r * VM3_REG_DW;
#pragma unroll
    for (int i = 0; i < 13; i++) x.l[i] = p[i];
    return x;
}
ECG_HD void vm3_store(const Vm3Regs& R, u32 r, const Fp& x) {
    u32* p = R.own + r * VM3_REG_DW;
#pragma unroll
    for (int i = 0; i < 13; i++) p[i] = x.l[i];
}

// c_own * own + c_par * par + K p with small signed coefficients; the result is positive by construction (K p covers the
// negative terms) and below 2^9 p; limbs renormalised to 30 bits except the top one
ECG_HD Fp vm3_derive(const Fp& own, const Fp& par, int c_own, int c_par, u32 k) {
    Fp s;
    int64_t cy = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        // (round 6, last) a limb is < 2^31 -- <= 2^30 + a few below the top, < 2^9 * 2^21 at the top -- so BOTH factors are stated as signed
        // 32-bit: one v_mad_i64_i32 per term.  Zero-extending the limb made each a 32 x 33-bit signed product = five instructions
        // (the same finding as bls_row.h rv_mad64s): a derived output 224 -> 150 instructions, a lone lane-group check 3.4 -> 3.1 ms.
        const int64_t t = (int64_t)c_own * (int64_t)(int32_t)own.l[i] + (int64_t)c_par * (int64_t)(int32_t)par.l[i] + (int64_t)k * (int64_t)blsc::P[i] + cy;
        s.l[i] = i + 1 < FP_N ? (u32)((u64)t & FP_MASK) : (u32)t;
        cy = t >> 30;
    }
    return s;
}

template <int N>
ECG_HD Fp vm3_sum(const Vm3Regs& R, const u32* w) {
    Fp a[N], b[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int ia = 1 + k;  // byte index in (w0, w1)
        a[k] = vm3_load(R, (w[ia >> 2] >> ((ia & 3) * 8)) & 255);
        b[k] = vm3_load(R, (w[2 + (k >> 2)] >> ((k & 3) * 8)) & 255);
    }
    return fp_sumprod<N>(a, b);
}

// the arithmetic of one lane in one round of class n (wave-uniform): its own result.  Three compiled sums -- 3, 4 and 7 products; round 2 / 3: 4 and 7 -- (the generator pads a
// round of N products to the next class: unused slots multiply ZERO by ZERO): together with the interpreter loop they are the
// whole hot code, ~50 KB -- inside the 64 KB instruction cache.
ECG_HD Fp vm3_own(u32 n, const Vm3Regs& R, const u32* w) {
    if (n == 0) return vm3_load(R, (w[0] >> 8) & 255);
    if (n <= 3) return vm3_sum<3>(R, w);
    if (n <= 4) return vm3_sum<4>(R, w);
    return vm3_sum<7>(R, w);
}

// Sequential (one tuple) execution with the lock-step semantics of the kernel: every lane of a round reads the register
// file as it was before the round.  Used by tests/hostsim.
inline void vm3_run_serial(const u32* prog, const u32* hdr, u32 rounds, u32 lanes, const Vm3Regs& R) {
    Fp own[64];
    for (u32 r = 0; r < rounds; r++) {
        const u32 n = hdr[r] & 255, nder = (hdr[r] >> 8) & 255;
        for (u32 k = 0; k < lanes; k++) own[k] = vm3_own(n, R, prog + ((size_t)r * lanes + k) * VM3_DESC_DW);
        // derived values are computed from the pre-round state too: collect, then write
        Fp der[64][4];
        for (u32 k = 0; k < lanes; k++) {
            const u32* w = prog + ((size_t)r * lanes + k) * VM3_DESC_DW;
            for (u32 d = 0; d < nder; d++) {
                const u32 x = w[4 + d];
                der[k][d] = vm3_derive(own[k], own[k ^ 1], (int)(int8_t)(x >> 8), (int)(int8_t)(x >> 16), x >> 24);
            }
        }
        for (u32 k = 0; k < lanes; k++) {
            const u32* w = prog + ((size_t)r * lanes + k) * VM3_DESC_DW;
            if (n && (w[0] & 255)) vm3_store(R, w[0] & 255, own[k]);
            for (u32 d = 0; d < nder; d++)
                if (w[4 + d] & 255) vm3_store(R, w[4 + d] & 255, der[k][d]);
        }
    }
}

}  // namespace ecg
