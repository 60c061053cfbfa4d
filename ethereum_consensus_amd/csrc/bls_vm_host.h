// Host-side interface of the sum-of-products lane-group pairing kernels (bls_vm3.hip) used by bls.hip.
#pragma once
#include "bls_kernels.h"
#include "runtime.h"

namespace ecg {

// Enqueue on `s`: for every tuple i < n the status of fast_aggregate_verify given the staged results (aggregate key, H(m),
// decoded signature and their statuses).  xfer: vm3_xfer_bytes(n) of workspace.
// a generated program (tools/gen_bls_vm3.py) in device memory
struct Vm3Desc {
    const u32* prog;       // rounds x LANES x 8 descriptor dwords
    const u32* hdr;        // rounds header words
    const u32* const_reg;  // nconst register numbers
    const u32* const_val;  // nconst x 13 limbs (Montgomery)
    u32 rounds, nreg, nconst, nin, nout;
    u32 in_reg[16], out_reg[16];
};
const Vm3Desc& vm3_program(int part);  // 0: Miller loops (16 lane slots), 1: final exponentiation (12); after init_vm3_tables
void vm3_launch_inv(hipStream_t s, u32* xfer, u32 n);  // the Fp inversion between the two parts, one lane per tuple
constexpr int VM3_SLOTS_A = 16, VM3_SLOTS_C = 12;  // lane slots of the two programs (asserted against the generated header in bls_vm3.hip)
constexpr u32 VM3_XFER_REGS = 16, VM3_XFER_STRIDE = VM3_XFER_REGS * 13;  // dwords per tuple between the parts
int init_vm3_tables();
size_t vm3_xfer_bytes(u32 n);
// the same check with the programs executed by the ROW machine (bls_row.hip): one workgroup per tuple, one Fp operation per
// 16-lane row -- the latency path of small batches
int row_programs();  // the row machine's copy of the two programs on the calling thread's device (built once per device: at its initialisation)
int row_pairing_launch(hipStream_t s, const A1* agg, const u8* st_pk, const u32* pk_off, const A2* hpts, const A2* sigpts, const u8* st_dec,
                       const u8* st_grp, const u8* sigs96, u32 n, int eth_variant, u8* d_status, u32* xfer);
int vm3_pairing_launch(hipStream_t s, const A1* agg, const u8* st_pk, const u32* pk_off, const A2* hpts, const A2* sigpts, const u8* st_dec,
                       const u8* st_grp, const u8* sigs96, u32 n, int eth_variant, u8* d_status, u32* xfer);

}  // namespace ecg
