// Host-side interface of the sum-of-products lane-group pairing kernels (bls_vm3.hip) used by bls.hip.
#pragma once
#include "bls_kernels.h"
#include "runtime.h"

namespace ecg {

// Enqueue on `s`: for every tuple i < n the status of fast_aggregate_verify given the staged results (aggregate key, H(m),
// decoded signature and their statuses).  xfer: vm3_xfer_bytes(n) of workspace.
int init_vm3_tables();
size_t vm3_xfer_bytes(u32 n);
int vm3_pairing_launch(hipStream_t s, const A1* agg, const u8* st_pk, const u32* pk_off, const A2* hpts, const A2* sigpts, const u8* st_dec,
                       const u8* st_grp, const u8* sigs96, u32 n, int eth_variant, u8* d_status, u32* xfer);

}  // namespace ecg
