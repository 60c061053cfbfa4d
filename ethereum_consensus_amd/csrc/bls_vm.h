// The "field VM": lane-group execution of straight-line Fp programs with an LDS-resident register
// file (see tools/gen_bls_vm.py for the why and for the programs).
//
// A tuple (one pairing check) is owned by ECG_VM_LANES consecutive lanes of a wave.  A program is a
// sequence of rounds; round r gives lane slot k the word prog[r * LANES + k]:
//     op[31:30] (0 nop, 1 mul, 2 add, 3 sub)   dst[29:20]   a[19:10]   b[9:0]
// All active slots of a round carry the same op, so a wave executes one Fp product (or one Fp
// addition) per round with every lane on its own operands: no divergence, no private memory.
// Registers are Fp values (13 dwords) in the tuple's slice of LDS; a register is never reused in
// the round that last reads it, so the lock-step read-then-write of a round needs no extra barrier.
#pragma once
#include "bls_fp.h"

namespace ecg {

ECG_HD Fp vm_load(const u32* R, u32 r) {
    Fp x;
    const u32* p = R + r * 13;
#pragma unroll
    for (int i = 0; i < 13; i++) x.l[i] = p[i];
    return x;
}
ECG_HD void vm_store(u32* R, u32 r, const Fp& x) {
    u32* p = R + r * 13;
#pragma unroll
    for (int i = 0; i < 13; i++) p[i] = x.l[i];
}

// One slot of one round: returns false for a nop.
ECG_HD bool vm_slot(u32 ins, const u32* R, Fp& out, u32& dst) {
    const u32 op = ins >> 30;
    if (op == 0) return false;
    dst = (ins >> 20) & 1023;
    Fp x = vm_load(R, (ins >> 10) & 1023);
    Fp y = vm_load(R, ins & 1023);
    if (op == 1) {
        out = fp_mul(x, y);
    } else if (op == 2) {
        out = fp_add(x, y);
    } else {
        out = fp_sub(x, y);
    }
    return true;
}

// Sequential (one tuple) execution with the lock-step semantics of the GPU kernel: every slot of a
// round reads the register file as it was before the round.  Used by tests/hostsim.
inline void vm_run_serial(const u32* prog, u32 rounds, u32 lanes, u32* R) {
    Fp res[64];
    u32 dst[64];
    bool act[64];
    for (u32 r = 0; r < rounds; r++) {
        for (u32 k = 0; k < lanes; k++) act[k] = vm_slot(prog[r * lanes + k], R, res[k], dst[k]);
        for (u32 k = 0; k < lanes; k++)
            if (act[k]) vm_store(R, dst[k], res[k]);
    }
}

}  // namespace ecg
