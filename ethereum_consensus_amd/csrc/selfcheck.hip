// Box self-check: does instruction fetch keep up with straight-line code that is much larger than the instruction cache?
//
// The BLS lane kernels are megabytes of straight-line multiply-adds at one wave per SIMD.  On most boxes of the pool a
// loop of 1 MB of code runs as fast per instruction as a loop of 8 KB; on a minority it runs several times slower (and
// so do k_pairing, k_sig, k_h2c -- profiles/r01zg_bls_probe_slow_box.txt -- while small-code kernels are unaffected).
// bench.py reports the two timings next to its numbers so that a reader can tell a slow kernel from a slow box.
#include "runtime.h"

namespace ecg {

// the probe kernels live in selfcheck_kernels.hip (1 MB of straight-line code takes minutes to compile; this file is the
// host side and may change with the ABI)
void launch_ifetch_probe(int mads, hipStream_t s, u32* out, u32 trips);

template <int MADS>
static int time_probe(hipStream_t s, u32* d_out, u32 trips, double* ms) {
    hipEvent_t e0, e1;
    ECG_HIP_CHECK(hipEventCreate(&e0));
    ECG_HIP_CHECK(hipEventCreate(&e1));
    launch_ifetch_probe(MADS, s, d_out, trips);  // warm-up: code load, caches
    ECG_HIP_CHECK(hipEventRecord(e0, s));
    launch_ifetch_probe(MADS, s, d_out, trips);
    ECG_HIP_CHECK(hipEventRecord(e1, s));
    ECG_HIP_CHECK(hipEventSynchronize(e1));
    float f = 0;
    ECG_HIP_CHECK(hipEventElapsedTime(&f, e0, e1));
    *ms = f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ECGPU_SUCCESS;
}

}  // namespace ecg

using namespace ecg;

extern "C" int ecgpu_selfcheck_ifetch(double* ms_small_loop, double* ms_large_loop) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!ms_small_loop || !ms_large_loop) return ECGPU_ERR_BAD_ARG;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    Arena& ar = c->arena(s);
    ar.reset();
    rc = ar.reserve(1024 * 64 * 4 + 256);
    if (rc) return rc;
    u32* d_out = (u32*)ar.take(1024 * 64 * 4);
    if (!d_out) return ECGPU_ERR_OOM;
    // the same 2^21 multiply-adds per lane: 2048 trips through 8 KB of code, 16 trips through 1 MB
    rc = time_probe<1024>(s, d_out, 2048, ms_small_loop);
    if (rc) return rc;
    return time_probe<131072>(s, d_out, 16, ms_large_loop);
}

// the same work over loops of 8 KB, 64 KB, 256 KB and 1 MB of code: where the slowdown of a slow box sets in
extern "C" int ecgpu_selfcheck_ifetch_sweep(double ms[4]) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!ms) return ECGPU_ERR_BAD_ARG;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    Arena& ar = c->arena(s);
    ar.reset();
    rc = ar.reserve(1024 * 64 * 4 + 256);
    if (rc) return rc;
    u32* d_out = (u32*)ar.take(1024 * 64 * 4);
    if (!d_out) return ECGPU_ERR_OOM;
    if ((rc = time_probe<1024>(s, d_out, 2048, &ms[0]))) return rc;
    if ((rc = time_probe<8192>(s, d_out, 256, &ms[1]))) return rc;
    if ((rc = time_probe<32768>(s, d_out, 64, &ms[2]))) return rc;
    return time_probe<131072>(s, d_out, 16, &ms[3]);
}
