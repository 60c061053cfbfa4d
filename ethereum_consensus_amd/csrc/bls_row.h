// The ROW machine: one Fp operation across a 16-lane row, limb j of every operand in lane j (BASELINE north_star: "384-bit limbs
// staged in LDS and carries reduced via wavefront shuffles").  It executes the SAME generated programs as the lane groups
// (tools/gen_bls_vm3.py, csrc/bls_vm3.h: rounds of sums of <= 7 products over an LDS-resident Fp register file, linear steps as
// derived outputs) -- but where a lane group gives each of a round's 16 sums to ONE lane (1 981 dependent instructions for seven
// products), the row machine gives it to a ROW of 16 lanes:
//   * an operand's limb j is read by lane j (a-side) and its 13 limbs by every lane (b-side: four LDS reads of a 16-byte aligned
//     register image, broadcast within the row);
//   * 13 iterations of { N multiply-adds a_j * b_i into ONE 64-bit accumulator per lane; the quotient digit from lane 0's low
//     dword (DPP row broadcast); one multiply-add by p_j; the window moves one limb down: every lane hands its low 30 bits to the
//     lane below (DPP row shift) and keeps its high part } -- 13 (N + 8) instructions, 195 for seven products;
//   * two carry passes leave limbs <= 2^30 (a lazy representation every product accepts; the value is the same integer < 2p the
//     one-lane sum returns);
//   * a derived output c_own * own + c_par * partner + K p takes the partner's limb from the adjacent row of the same wave and
//     resolves its signed carries with a bias that cancels (2^40 into limb j, -2^10 out of limb j + 1).
// A tuple is one WORKGROUP: 16 (Miller loops) / 12 (final exponentiation) rows = 4 / 3 waves, one per SIMD of a CU, two
// barriers per round.  This is the latency path of the pairing check: a round costs ~0.5 us instead of ~4.2, a lone
// verification's pairing ~0.6 ms instead of 3.45.  The crate's callers verify ONE signature per call
// (/root/reference/ethereum-consensus/src/crypto/bls.rs:64-77, phase0/block_processing.rs:753-761, altair/block_processing.rs:230).
//
// Written once over a small lane-vector vocabulary (rv_*): on the device a value is one lane's dword and the primitives are DPP /
// ds_bpermute / LDS instructions; on the host (tests/hostsim) a value is an array over the 32 lanes of a row PAIR executed in
// lock step, so the CPU test-suite runs the very same routines.
#pragma once
#include "bls_vm3.h"

namespace ecg {

constexpr u32 ROW_REG_DW = 16;  // dwords per register image in the row machine's LDS file: 13 limbs + 3 zero words, 16-byte aligned

#if defined(__HIPCC__)
typedef u32 rv32;
typedef u64 rv64;
#define ROW_FN __device__ __forceinline__
ROW_FN rv32 rv_splat(u32 x) { return x; }
ROW_FN rv32 rv_lane() { return threadIdx.x & 15u; }
ROW_FN rv32 rv_and(rv32 a, rv32 b) { return a & b; }
ROW_FN rv32 rv_add(rv32 a, rv32 b) { return a + b; }
ROW_FN rv32 rv_sub(rv32 a, rv32 b) { return a - b; }
ROW_FN rv32 rv_mul_lo(rv32 a, rv32 b) { return a * b; }
ROW_FN rv32 rv_shr(rv32 a, u32 n) { return a >> n; }
ROW_FN rv32 rv_byte(rv32 a, u32 k) { return (a >> (8 * k)) & 255u; }
ROW_FN rv32 rv_lt(rv32 a, u32 k) { return a < k ? 1u : 0u; }
ROW_FN rv32 rv_sel(rv32 c, rv32 a, rv32 b) { return c ? a : b; }
ROW_FN rv64 rv_zero64() { return 0; }
ROW_FN rv64 rv_mad64(rv32 a, rv32 b, rv64 c) { return (u64)a * b + c; }  // v_mad_u64_u32
// b < 2^31 (a limb <= 2^30 + a few, or the small non-negative top limb of a value in range): BOTH operands are stated as signed
// 32-bit, which is what makes this ONE v_mad_i64_i32.  (Round 5 extended b as unsigned: the compiler then emulates the 32 x 33-bit
// signed product with two v_mad_u64_u32, an arithmetic shift and two moves -- five instructions per term of every linear step,
// 12 of the ~45 instructions of an addition or subtraction on a row: profiles/r06e_row_linear_ops.txt.)
ROW_FN rv64 rv_mad64s(rv32 a_signed, rv32 b, rv64 c) { return (u64)((int64_t)(int32_t)a_signed * (int64_t)(int32_t)b + (int64_t)c); }
ROW_FN rv32 rv_lo(rv64 a) { return (u32)a; }
ROW_FN rv64 rv_shr64(rv64 a, u32 n) { return a >> n; }
ROW_FN rv64 rv_sar64(rv64 a, u32 n) { return (u64)((int64_t)a >> n); }
ROW_FN rv64 rv_add64(rv64 a, rv32 b) { return a + b; }
ROW_FN rv64 rv_add64c(rv64 a, u64 k) { return a + k; }
ROW_FN rv64 rv_sel64(rv32 c, rv64 a, rv64 b) { return c ? a : b; }
ROW_FN rv32 rv_sext8(rv32 a) { return (u32)(int32_t)(int8_t)a; }
// lane 0 of the lane's own row
ROW_FN rv32 rv_bcast0(rv32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x150, 0xf, 0xf, true); }  // row_newbcast:0
// lane j <- lane j + 1 of the same row (lane 15 <- 0) / lane j <- lane j - 1 (lane 0 <- 0)
ROW_FN rv32 rv_from_next(rv32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xf, 0xf, true); }  // row_shl:1
ROW_FN rv32 rv_from_prev(rv32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true); }  // row_shr:1
// the same lane of the other row of the pair (rows 2m, 2m + 1 of a wave)
ROW_FN rv32 rv_partner(rv32 x) { return (u32)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63u) ^ 16u) << 2), (int)x); }
// the two row PAIRS of a wave (lanes 0 .. 31, 32 .. 63: bls_rowcurve.h jac_dbl_quad): which one the lane is in, and the same lane
// of the other one
ROW_FN rv32 rv_quad_hi() { return (threadIdx.x >> 5) & 1u; }
ROW_FN rv32 rv_other_pair(rv32 x) { return (u32)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63u) ^ 32u) << 2), (int)x); }
// lane I of the lane's own row
template <int I>
ROW_FN rv32 rv_bcast(rv32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 + I, 0xf, 0xf, true); }  // row_newbcast:I
ROW_FN rv32 rv_or(rv32 a, rv32 b) { return a | b; }
ROW_FN rv32 rv_xor(rv32 a, rv32 b) { return a ^ b; }
ROW_FN rv32 rv_shl(rv32 a, u32 n) { return a << n; }
// the value in a vector register the compiler cannot see through (so that it stays one: see row_sumprod)
ROW_FN rv32 rv_in_vgpr(rv32 x) {
    asm volatile("" : "+v"(x));
    return x;
}
ROW_FN rv32 rv_eq(rv32 a, rv32 b) { return a == b ? 1u : 0u; }
ROW_FN rv32 rv_sar(rv32 a, u32 n) { return (u32)((int32_t)a >> n); }
// 1 in every lane of the row if x != 0 in any of its lanes (the rows of a wave decide independently)
ROW_FN rv32 rv_row_any(rv32 x) {
    const u64 m = __builtin_amdgcn_ballot_w64(x != 0);
    return ((m >> (threadIdx.x & 48u)) & 0xffffull) != 0 ? 1u : 0u;
}
ROW_FN bool rv_test(rv32 c) { return c != 0; }  // a row-uniform condition as a branch condition
// a row PAIR (rows 2m, 2m + 1 of a wave: the two components of an Fp2 value, bls_rowpair.h): which row the lane is in, and
// "any lane of the pair"
ROW_FN rv32 rv_pair_row() { return (threadIdx.x >> 4) & 1u; }
ROW_FN rv32 rv_pair_any(rv32 x) {
    const u64 m = __builtin_amdgcn_ballot_w64(x != 0);
    return ((m >> (threadIdx.x & 32u)) & 0xffffffffull) != 0 ? 1u : 0u;
}
// exact carry propagation over limbs <= 2^30: a limb == 2^30 generates, a limb == 2^30 - 1 propagates; the chain is resolved by
// ONE 64-bit addition of the two ballots (rows end in zero limbs, which neither generate nor propagate)
// (limbs 0 .. 11 come out < 2^30; the top limb, lane 12, takes its carry and keeps every bit -- it may be a signed dword)
ROW_FN rv32 rv_carry_exact(rv32 v) {
    const bool low = (threadIdx.x & 15u) < 12;
    // (the lane condition as a constant mask on the scalar side: a ballot of `low && ...` costs a select and a second compare per ballot)
    constexpr u64 LOW_LANES = 0x0fff0fff0fff0fffull;
    const u64 g = __builtin_amdgcn_ballot_w64(v > FP_MASK) & LOW_LANES, pr = __builtin_amdgcn_ballot_w64(v == FP_MASK) & LOW_LANES;
    const u64 y = g << 1, cin = ((pr + y) ^ pr ^ y) | y;
    const u32 t = v + (u32)((cin >> (threadIdx.x & 63u)) & 1ull);
    return low ? (t & FP_MASK) : t;
}
ROW_FN rv32 rv_lds_read(const u32* lds, rv32 dw) { return lds[dw]; }
ROW_FN void rv_lds_read4(const u32* lds, rv32 dw, rv32* out) {  // 16-byte aligned
    const uint4 q = *reinterpret_cast<const uint4*>(lds + dw);
    out[0] = q.x, out[1] = q.y, out[2] = q.z, out[3] = q.w;
}
ROW_FN void rv_lds_write(u32* lds, rv32 dw, rv32 v, rv32 pred) {
    if (pred) lds[dw] = v;
}
#else
#ifndef ECG_ROW_SIM
#define ECG_ROW_SIM 32  // host: the 32 lanes of one row pair in lock step (tests/hostsim/hostsim_quad.cpp: 64, the two pairs of a wave)
#endif
constexpr int ROW_SIM = ECG_ROW_SIM;
struct rv32 {
    u32 v[ROW_SIM];
};
struct rv64 {
    u64 v[ROW_SIM];
};
#define ROW_FN static inline  // (static: the simulator has translation units with different lane-vector widths)
#define ROW_EACH for (int l_ = 0; l_ < ROW_SIM; l_++)
ROW_FN rv32 rv_splat(u32 x) { rv32 r; ROW_EACH r.v[l_] = x; return r; }
ROW_FN rv32 rv_lane() { rv32 r; ROW_EACH r.v[l_] = l_ & 15; return r; }
ROW_FN rv32 rv_and(rv32 a, rv32 b) { rv32 r; ROW_EACH r.v[l_] = a.v[l_] & b.v[l_]; return r; }
ROW_FN rv32 rv_add(rv32 a, rv32 b) { rv32 r; ROW_EACH r.v[l_] = a.v[l_] + b.v[l_]; return r; }
ROW_FN rv32 rv_sub(rv32 a, rv32 b) { rv32 r; ROW_EACH r.v[l_] = a.v[l_] - b.v[l_]; return r; }
ROW_FN rv32 rv_mul_lo(rv32 a, rv32 b) { rv32 r; ROW_EACH r.v[l_] = a.v[l_] * b.v[l_]; return r; }
ROW_FN rv32 rv_shr(rv32 a, u32 n) { rv32 r; ROW_EACH r.v[l_] = a.v[l_] >> n; return r; }
ROW_FN rv32 rv_byte(rv32 a, u32 k) { rv32 r; ROW_EACH r.v[l_] = (a.v[l_] >> (8 * k)) & 255u; return r; }
ROW_FN rv32 rv_lt(rv32 a, u32 k) { rv32 r; ROW_EACH r.v[l_] = a.v[l_] < k ? 1u : 0u; return r; }
ROW_FN rv32 rv_sel(rv32 c, rv32 a, rv32 b) { rv32 r; ROW_EACH r.v[l_] = c.v[l_] ? a.v[l_] : b.v[l_]; return r; }
ROW_FN rv64 rv_zero64() { rv64 r; ROW_EACH r.v[l_] = 0; return r; }
extern unsigned long long g_ecg_column_overflows;
ROW_FN rv64 rv_mad64(rv32 a, rv32 b, rv64 c) {
    rv64 r;
    ROW_EACH if (__builtin_add_overflow((u64)a.v[l_] * b.v[l_], c.v[l_], &r.v[l_])) g_ecg_column_overflows++;
    return r;
}
ROW_FN rv64 rv_mad64s(rv32 a, rv32 b, rv64 c) {
    rv64 r;
    ROW_EACH r.v[l_] = (u64)((int64_t)(int32_t)a.v[l_] * (int64_t)b.v[l_] + (int64_t)c.v[l_]);
    return r;
}
ROW_FN rv32 rv_lo(rv64 a) { rv32 r; ROW_EACH r.v[l_] = (u32)a.v[l_]; return r; }
ROW_FN rv64 rv_shr64(rv64 a, u32 n) { rv64 r; ROW_EACH r.v[l_] = a.v[l_] >> n; return r; }
ROW_FN rv64 rv_sar64(rv64 a, u32 n) { rv64 r; ROW_EACH r.v[l_] = (u64)((int64_t)a.v[l_] >> n); return r; }
ROW_FN rv64 rv_add64(rv64 a, rv32 b) { rv64 r; ROW_EACH r.v[l_] = a.v[l_] + b.v[l_]; return r; }
ROW_FN rv64 rv_add64c(rv64 a, u64 k) { rv64 r; ROW_EACH r.v[l_] = a.v[l_] + k; return r; }
ROW_FN rv64 rv_sel64(rv32 c, rv64 a, rv64 b) { rv64 r; ROW_EACH r.v[l_] = c.v[l_] ? a.v[l_] : b.v[l_]; return r; }
ROW_FN rv32 rv_sext8(rv32 a) { rv32 r; ROW_EACH r.v[l_] = (u32)(int32_t)(int8_t)a.v[l_]; return r; }
ROW_FN rv32 rv_bcast0(rv32 x) { rv32 r; ROW_EACH r.v[l_] = x.v[l_ & ~15]; return r; }
ROW_FN rv32 rv_from_next(rv32 x) { rv32 r; ROW_EACH r.v[l_] = (l_ & 15) == 15 ? 0u : x.v[l_ + 1]; return r; }
ROW_FN rv32 rv_from_prev(rv32 x) { rv32 r; ROW_EACH r.v[l_] = (l_ & 15) == 0 ? 0u : x.v[l_ - 1]; return r; }
ROW_FN rv32 rv_partner(rv32 x) { rv32 r; ROW_EACH r.v[l_] = x.v[l_ ^ 16]; return r; }
ROW_FN rv32 rv_quad_hi() { rv32 r; ROW_EACH r.v[l_] = (l_ >> 5) & 1; return r; }
ROW_FN rv32 rv_other_pair(rv32 x) { rv32 r; ROW_EACH r.v[l_] = x.v[(l_ ^ 32) % ROW_SIM]; return r; }
template <int I>
ROW_FN rv32 rv_bcast(rv32 x) { rv32 r; ROW_EACH r.v[l_] = x.v[(l_ & ~15) + I]; return r; }
ROW_FN rv32 rv_or(rv32 a, rv32 b) { rv32 r; ROW_EACH r.v[l_] = a.v[l_] | b.v[l_]; return r; }
ROW_FN rv32 rv_xor(rv32 a, rv32 b) { rv32 r; ROW_EACH r.v[l_] = a.v[l_] ^ b.v[l_]; return r; }
ROW_FN rv32 rv_shl(rv32 a, u32 n) { rv32 r; ROW_EACH r.v[l_] = a.v[l_] << n; return r; }
ROW_FN rv32 rv_in_vgpr(rv32 x) { return x; }
ROW_FN rv32 rv_eq(rv32 a, rv32 b) { rv32 r; ROW_EACH r.v[l_] = a.v[l_] == b.v[l_] ? 1u : 0u; return r; }
ROW_FN rv32 rv_sar(rv32 a, u32 n) { rv32 r; ROW_EACH r.v[l_] = (u32)((int32_t)a.v[l_] >> n); return r; }
ROW_FN rv32 rv_row_any(rv32 x) {
    rv32 r;
    ROW_EACH {
        u32 any = 0;
        for (int q = 0; q < 16; q++) any |= x.v[(l_ & ~15) + q];
        r.v[l_] = any ? 1u : 0u;
    }
    return r;
}
// (the host runs the SAME computation on both simulated rows wherever a condition steers control flow: lane 0 speaks for both)
ROW_FN bool rv_test(rv32 c) { return c.v[0] != 0; }
ROW_FN rv32 rv_pair_row() { rv32 r; ROW_EACH r.v[l_] = (l_ >> 4) & 1; return r; }
ROW_FN rv32 rv_pair_any(rv32 x) {
    rv32 r;
    for (int pair = 0; pair < ROW_SIM; pair += 32) {
        u32 any = 0;
        for (int q = 0; q < 32; q++) any |= x.v[pair + q];
        for (int q = 0; q < 32; q++) r.v[pair + q] = any ? 1u : 0u;
    }
    return r;
}
ROW_FN rv32 rv_carry_exact(rv32 v) {
    rv32 r;
    for (int row = 0; row < ROW_SIM; row += 16) {
        u32 c = 0;
        for (int q = 0; q < 16; q++) {
            const u32 t = v.v[row + q] + c;
            r.v[row + q] = q < 12 ? (t & FP_MASK) : t;
            c = q < 12 ? t >> 30 : 0;
        }
    }
    return r;
}
ROW_FN rv32 rv_lds_read(const u32* lds, rv32 dw) { rv32 r; ROW_EACH r.v[l_] = lds[dw.v[l_]]; return r; }
ROW_FN void rv_lds_read4(const u32* lds, rv32 dw, rv32* out) {
    ROW_EACH for (int q = 0; q < 4; q++) out[q].v[l_] = lds[dw.v[l_] + q];
}
ROW_FN void rv_lds_write(u32* lds, rv32 dw, rv32 v, rv32 pred) {
    ROW_EACH if (pred.v[l_]) lds[dw.v[l_]] = v.v[l_];
}
#endif

// ---- the arithmetic of one row --------------------------------------------------------------------------------------------
// sum_{k < N} A_k * B_k / R mod p (one Montgomery reduction, result < 2p for sum of bound products < R / p): a[k] = limb
// `lane` of A_k, b[k][i] = limb i of B_k (the same in every lane of the row), p_limb = limb `lane` of p (0 beyond limb 12).
// Operand limbs <= 2^30; the accumulator of a lane never exceeds 2^35 + 8 * 2^60.
template <int N>
ROW_FN rv32 row_sumprod(const rv32 (&a)[N], const rv32 (&b)[N][13], rv32 p_limb) {
    const rv32 mask = rv_in_vgpr(rv_splat(FP_MASK)), n0 = rv_splat(blsc::N0);  // (a DPP instruction takes no literal operand)
    rv64 acc = rv_zero64();
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
#pragma unroll
        for (int k = 0; k < N; k++) acc = rv_mad64(a[k], b[k][i], acc);
        // the quotient digit: lane 0's column becomes 0 mod 2^30.  (Every lane multiplies its own column, lane 0's product is
        // the one broadcast -- in this order the broadcast and the shift below are DPP operands of the two v_and_b32, not
        // v_mov_b32_dpp of their own in front of a VOP3 instruction: 2 of the 12 + N instructions of an iteration.  The first
        // form of round 5 -- bcast(lo) * n0 & mask, from_next(lo & mask) -- and a form with the accumulator pinned after every
        // step were measured beside this one: profiles/r05w_*.)
        const rv32 m = rv_and(rv_bcast0(rv_mul_lo(rv_lo(acc), n0)), mask);
        acc = rv_mad64(m, p_limb, acc);
        // the window moves one limb down: lane j keeps its carry (column i + j + 1) and takes the low 30 bits of lane j + 1
        const rv64 hi = rv_shr64(acc, 30);
        acc = rv_add64(hi, rv_and(rv_from_next(rv_lo(acc)), mask));
    }
    // acc < 2^35: two carry passes upwards leave limbs <= 2^30 (the top limb of a value < 2p is far below)
    rv32 v = rv_add(rv_and(rv_lo(acc), mask), rv_from_prev(rv_lo(rv_shr64(acc, 30))));
    v = rv_add(rv_and(v, mask), rv_from_prev(rv_shr(v, 30)));
    return v;
}
// (Round 6 experiment, not kept: the compiler computes the 13 column sums first and then runs the reduction as one dependent chain;
// issuing the products of column i + 2 INSIDE iteration i -- a scheduling barrier per iteration held them there, the hazard slots
// shrank from two wait states to one -- made a lone check 1.01 -> 0.985 ms and 1 024 tuples 2.09 -> 2.36 ms: the interleaved form
// holds more accumulators live and the waves that share a SIMD lose more than the lone wave gains.  profiles/r06g_interleave_probe.txt)

// c_own * own + c_par * par + K p, small signed coefficients (the lane groups' vm3_derive): positive by construction, limbs
// <= 2^30 except the top one (limb 12), which keeps everything above it.
ROW_FN rv32 row_derive(rv32 own, rv32 par, rv32 c_own, rv32 c_par, rv32 k, rv32 p_limb, rv32 lane) {
    rv64 t = rv_mad64s(c_own, own, rv_zero64());
    t = rv_mad64s(c_par, par, t);
    t = rv_mad64s(k, p_limb, t);
    // signed limbs: + 2^40 into limb j < 12 and - 2^10 out of limb j + 1 cancel, and make every split limb positive
    const rv32 below_top = rv_lt(lane, 12), has_lower = rv_and(rv_lt(rv_sub(lane, rv_splat(1)), 12), rv_splat(1));  // lanes 1 .. 12
    t = rv_sel64(below_top, rv_add64c(t, 1ull << 40), t);
    t = rv_sel64(has_lower, rv_add64c(t, (u64)0 - (1ull << 10)), t);
    const rv32 mask = rv_sel(below_top, rv_splat(FP_MASK), rv_splat(0xffffffffu)), zero = rv_splat(0);
    rv32 hi = rv_sel(below_top, rv_lo(rv_sar64(t, 30)), zero);
    rv32 v = rv_add(rv_and(rv_lo(t), mask), rv_from_prev(hi));
    hi = rv_sel(below_top, rv_shr(v, 30), zero);
    v = rv_add(rv_and(v, mask), rv_from_prev(hi));
    // exact limbs below the top: a limb == 2^30 left standing would let a value below 2^360 come out with top limb -1
    return rv_carry_exact(v);
}

// ---- one round of a program for one row ---------------------------------------------------------------------------------------
// The register file of the workgroup's tuple: registers [0, nreg) then the program's constants, ROW_REG_DW dwords each.
struct RowFile {
    u32* lds;
    u32 nreg;
};
// first dword of register r.  The kernels run a copy of the program whose register numbers ARE positions in the file
// (bls_row.hip row_programs: constants renumbered behind the registers); the host simulator runs the generator's numbering,
// where numbers from VM3_CONST_BASE name constants.
ROW_FN rv32 row_reg_dw(const RowFile& F, rv32 r) {
#if defined(__HIPCC__)
    return r << 4;
#else
    const rv32 is_own = rv_lt(r, VM3_CONST_BASE);
    return rv_mul_lo(rv_sel(is_own, r, rv_add(rv_sub(r, rv_splat(VM3_CONST_BASE)), rv_splat(F.nreg))), rv_splat(ROW_REG_DW));
#endif
}
struct RowResult {
    rv32 own, der[4];
};
template <int N>
ROW_FN rv32 row_round_sum(const RowFile& F, const rv32 (&w)[8], rv32 lane, rv32 p_limb) {
    rv32 a[N], b[N][13];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int ia = 1 + k;
        const rv32 ra = rv_byte(w[ia >> 2], ia & 3), rb = rv_byte(w[2 + (k >> 2)], k & 3);
        a[k] = rv_lds_read(F.lds, rv_add(row_reg_dw(F, ra), lane));
        const rv32 bdw = row_reg_dw(F, rb);
        rv32 q[16];
#pragma unroll
        for (int c = 0; c < 4; c++) rv_lds_read4(F.lds, rv_add(bdw, rv_splat(4 * c)), q + 4 * c);
#pragma unroll
        for (int i = 0; i < 13; i++) b[k][i] = q[i];
    }
    return row_sumprod<N>(a, b, p_limb);
}
// everything a row computes in a round, from the register file as it was before the round (n, nder: the round's header)
ROW_FN RowResult row_round_compute(const RowFile& F, u32 n, u32 nder, const rv32 (&w)[8], rv32 p_limb) {
    const rv32 lane = rv_lane();
    RowResult r;
    if (n == 0)
        r.own = rv_lds_read(F.lds, rv_add(row_reg_dw(F, rv_byte(w[0], 1)), lane));
    else if (n <= 3)
        r.own = row_round_sum<3>(F, w, lane, p_limb);
    else if (n <= 4)
        r.own = row_round_sum<4>(F, w, lane, p_limb);
    else
        r.own = row_round_sum<7>(F, w, lane, p_limb);
    if (nder) {
        const rv32 par = rv_partner(r.own);
#pragma unroll
        for (u32 d = 0; d < 4; d++) {  // (constant indices: the descriptor stays in registers)
            if (d >= nder) break;
            const rv32 x = w[4 + d];
            r.der[d] = row_derive(r.own, par, rv_sext8(rv_byte(x, 1)), rv_sext8(rv_byte(x, 2)), rv_byte(x, 3), p_limb, lane);
        }
    }
    return r;
}
// ... and its writes (after every row of the workgroup has read)
ROW_FN void row_round_store(const RowFile& F, u32 n, u32 nder, const rv32 (&w)[8], const RowResult& r) {
    const rv32 lane = rv_lane(), one = rv_splat(1);
    const rv32 dst = rv_byte(w[0], 0);
    if (n) rv_lds_write(F.lds, rv_add(row_reg_dw(F, dst), lane), r.own, rv_sub(one, rv_lt(dst, 1)));
#pragma unroll
    for (u32 d = 0; d < 4; d++) {
        if (d >= nder) break;
        const rv32 reg = rv_byte(w[4 + d], 0);
        rv_lds_write(F.lds, rv_add(row_reg_dw(F, reg), lane), r.der[d], rv_sub(one, rv_lt(reg, 1)));
    }
}

// an Fp (13 limbs, exact 30-bit limbs) out of a register image whose limbs may be 2^30
ECG_HD Fp row_image_to_fp(const u32* img) {
    Fp x;
    u32 cy = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        const u32 t = img[i] + cy;
        x.l[i] = i + 1 < FP_N ? (t & FP_MASK) : t;
        cy = t >> 30;
    }
    return x;
}

}  // namespace ecg
