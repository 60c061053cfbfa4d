// The "Fp2 VM": lane-group execution of straight-line Fp2 programs with an LDS-resident register
// file (tools/gen_bls_vm2.py has the why, the programs and the encoding).
//
// A tuple (one pairing check) is owned by ECG_VM2_LANES consecutive lanes of a wave.  A program is a
// sequence of rounds; round r gives lane slot k the word prog[r * LANES + k]:
//     op[31:28] flags[27:24] dst[23:16] a[15:8] b[7:0]
// and the wave one class byte cls[r] saying which variants occur in the round (wave-uniform branches
// skip the absent ones).  A round is either a PRODUCT round (Fp2 product / square / Fp scaling / norm:
// 3 or 2 Montgomery products per lane, Karatsuba additions in VGPRs) or a LINEAR round (a +- b with a
// sign per component, a +- xi b).  Registers are Fp2 values (26 dwords) in the tuple's slice of LDS;
// a register is never reused in the round that last reads it, so the lock-step read-then-write of a
// round needs no extra barrier.
#pragma once
#include "bls_fp.h"

namespace ecg {

enum : u32 { VM2_NOP = 0, VM2_MUL = 1, VM2_SQR = 2, VM2_MULFP = 3, VM2_NORM = 4, VM2_LIN = 5, VM2_LINXI = 6 };
enum : u32 { VM2C_MUL = 1, VM2C_SQR = 2, VM2C_MULFP = 4, VM2C_NORM = 8, VM2C_LIN = 16, VM2C_LINXI = 32, VM2C_PROD = 15 };
constexpr u32 VM2_REG_DW = 26;  // dwords per register

ECG_HD Fp2 vm2_load(const u32* R, u32 r) {
    Fp2 x;
    const u32* p = R + r * VM2_REG_DW;
#pragma unroll
    for (int i = 0; i < 13; i++) x.c0.l[i] = p[i];
#pragma unroll
    for (int i = 0; i < 13; i++) x.c1.l[i] = p[13 + i];
    return x;
}
ECG_HD void vm2_store(u32* R, u32 r, const Fp2& x) {
    u32* p = R + r * VM2_REG_DW;
#pragma unroll
    for (int i = 0; i < 13; i++) p[i] = x.c0.l[i];
#pragma unroll
    for (int i = 0; i < 13; i++) p[13 + i] = x.c1.l[i];
}

// One slot of one round; `cls` is the round's class byte.  Returns false for a nop.
// Every lane of the wave calls this in lock step; the Montgomery products are issued convergently
// (lanes whose operation has only two products idle through the third one of a MUL round).
ECG_HD bool vm2_slot(u32 ins, u32 cls, const u32* R, Fp2& out, u32& dst) {
    const u32 op = ins >> 28;
    if (op == VM2_NOP) return false;
    const u32 fl = (ins >> 24) & 15;
    dst = (ins >> 16) & 255;
    const Fp2 a = vm2_load(R, (ins >> 8) & 255);
    const Fp2 b = vm2_load(R, ins & 255);
    if (cls & VM2C_PROD) {
        Fp x0, y0, x1, y1, x2, y2;
        if (op == VM2_MUL) {
            x0 = a.c0, y0 = b.c0, x1 = a.c1, y1 = b.c1;
            x2 = fp_add(a.c0, a.c1), y2 = fp_add(b.c0, b.c1);
        } else if (op == VM2_SQR) {
            x0 = fp_add(a.c0, a.c1), y0 = fp_sub(a.c0, a.c1), x1 = a.c0, y1 = a.c1;
            x2 = y2 = a.c0;
        } else if (op == VM2_MULFP) {
            y0 = y1 = (fl & 1) ? b.c1 : b.c0;
            x0 = a.c0, x1 = a.c1;
            x2 = y2 = a.c0;
        } else {  // VM2_NORM
            x0 = y0 = a.c0, x1 = y1 = a.c1;
            x2 = y2 = a.c0;
        }
        const Fp p0 = fp_mul(x0, y0);
        const Fp p1 = fp_mul(x1, y1);
        Fp p2 = p0;
        if (cls & VM2C_MUL) p2 = fp_mul(x2, y2);
        if (op == VM2_MUL) {
            out.c0 = fp_sub(p0, p1);
            out.c1 = fp_sub(fp_sub(p2, p0), p1);
        } else if (op == VM2_SQR) {
            out.c0 = p0;
            out.c1 = fp_add(p1, p1);
        } else if (op == VM2_MULFP) {
            out.c0 = p0;
            out.c1 = p1;
        } else {
            out.c0 = fp_add(p0, p1);
            out.c1 = fp_zero();
        }
    } else if (op == VM2_LIN) {
        out.c0 = (fl & 1) ? fp_sub(a.c0, b.c0) : fp_add(a.c0, b.c0);
        out.c1 = (fl & 2) ? fp_sub(a.c1, b.c1) : fp_add(a.c1, b.c1);
    } else {  // VM2_LINXI: a +- (b0 - b1, b0 + b1)
        const Fp x0 = fp_sub(b.c0, b.c1), x1 = fp_add(b.c0, b.c1);
        if (fl & 1) {
            out.c0 = fp_sub(a.c0, x0);
            out.c1 = fp_sub(a.c1, x1);
        } else {
            out.c0 = fp_add(a.c0, x0);
            out.c1 = fp_add(a.c1, x1);
        }
    }
    return true;
}

// Sequential (one tuple) execution with the lock-step semantics of the GPU kernel: every slot of a
// round reads the register file as it was before the round.  Used by tests/hostsim.
inline void vm2_run_serial(const u32* prog, const u8* cls, u32 rounds, u32 lanes, u32* R) {
    Fp2 res[64];
    u32 dst[64];
    bool act[64];
    for (u32 r = 0; r < rounds; r++) {
        for (u32 k = 0; k < lanes; k++) act[k] = vm2_slot(prog[r * lanes + k], cls[r], R, res[k], dst[k]);
        for (u32 k = 0; k < lanes; k++)
            if (act[k]) vm2_store(R, dst[k], res[k]);
    }
}

}  // namespace ecg
