// BLS12-381 kernels + C ABI (pipeline under construction: entry points report a backend fault,
// never a verdict, until the kernels land).
#include "runtime.h"
namespace ecg {
int init_bls_tables(hipStream_t) { return ECGPU_SUCCESS; }
}  // namespace ecg
using namespace ecg;
#define ECG_UNIMPL() do { set_last_error("BLS path not built yet"); return ECGPU_ERR_HIP; } while (0)
extern "C" {
int ecgpu_verify(const uint8_t*, const uint8_t*, size_t, const uint8_t*) { ECG_UNIMPL(); }
int ecgpu_fast_aggregate_verify(const uint8_t*, uint32_t, const uint8_t*, size_t, const uint8_t*, int) { ECG_UNIMPL(); }
int ecgpu_aggregate_verify(const uint8_t*, uint32_t, const uint8_t*, const uint64_t*, uint32_t, const uint8_t*) { ECG_UNIMPL(); }
int ecgpu_aggregate_sigs(const uint8_t*, uint32_t, uint8_t*) { ECG_UNIMPL(); }
int ecgpu_aggregate_pks(const uint8_t*, uint32_t, uint8_t*) { ECG_UNIMPL(); }
int ecgpu_fast_aggregate_verify_batch(const uint8_t*, const uint32_t*, const uint8_t*, const uint8_t*, uint32_t, int, uint8_t*) { ECG_UNIMPL(); }
int ecgpu_fast_aggregate_verify_batch_dev(const uint8_t*, const uint32_t*, uint32_t, const uint8_t*, const uint8_t*, uint32_t, int, uint8_t*, ecgpu_stream_t) { ECG_UNIMPL(); }
}
