// hash_tree_root(BeaconState) for the deneb fork, driven from the state's SSZ encoding
// (plan: state_plan.h).  The host only walks SSZ offsets and emits descriptors; every hash64
// runs on the GPU.  Big fields go through the pass kernels straight from the device-resident
// encoding (read once); the ~60 small chunks are gathered into one staging buffer and reduced
// by three batched k_tree_jobs launches (leaf containers -> nested containers -> the 28-field
// state container).
#include <cstring>
#include <vector>

#include "merkle_driver.h"
#include "state_plan.h"

namespace ecg {

static int state_root_device(hipStream_t s, ThreadCtx* c, const u8* d_ssz, u64 n_bytes, const u8* h_fixed,
                             int preset, u8* d_root) {
    StatePlan plan;
    if (!build_state_plan_deneb(h_fixed, n_bytes, preset, plan)) {
        set_last_error(plan.error);
        return ECGPU_ERR_BAD_ARG;
    }
    // ---- device buffers --------------------------------------------------------------------------
    Arena& ar = c->arena(s);
    ar.reset();
    size_t need = 4096;
    for (auto& b : plan.bigs) need += merkle_ws_bytes(b.n0) + 512;
    const size_t small_bytes = 32ull * plan.n_small_chunks;
    const size_t n_jobs = plan.jobs[0].size() + plan.jobs[1].size() + plan.jobs[2].size();
    need += small_bytes + n_jobs * sizeof(TreeJob) + plan.gathers.size() * sizeof(GatherDesc) + 2048;
    int rc = ar.reserve(need);
    if (rc) return rc;
    u8* d_small = ar.take(small_bytes);
    TreeJob* d_jobs = (TreeJob*)ar.take(n_jobs * sizeof(TreeJob));
    GatherDesc* d_gath = (GatherDesc*)ar.take(plan.gathers.size() * sizeof(GatherDesc));
    // descriptors travel through pageable memory: hipMemcpyAsync stages them before returning,
    // so the host vectors may die at the end of this call while the stream is still running.
    std::vector<TreeJob> all_jobs;
    for (int l = 0; l < 3; l++) all_jobs.insert(all_jobs.end(), plan.jobs[l].begin(), plan.jobs[l].end());
    ECG_HIP_CHECK(hipMemcpyAsync(d_jobs, all_jobs.data(), n_jobs * sizeof(TreeJob), hipMemcpyHostToDevice, s));
    ECG_HIP_CHECK(hipMemcpyAsync(d_gath, plan.gathers.data(), plan.gathers.size() * sizeof(GatherDesc),
                                 hipMemcpyHostToDevice, s));
    ECG_HIP_CHECK(hipMemsetAsync(d_small, 0, small_bytes, s));
    rc = launch_gather(s, d_ssz, n_bytes, d_gath, (u32)plan.gathers.size(), d_small);
    if (rc) return rc;
    u64 hc = plan.small_hashes;
    for (auto& b : plan.bigs) {
        u8* ws = ar.take(merkle_ws_bytes(b.n0));
        rc = merkleize_device(s, b.kind, d_ssz + b.src, b.bytes, b.n0, b.depth, b.mix, b.mix_len,
                              d_small + 32ull * b.out_chunk, ws, &hc);
        if (rc) return rc;
    }
    size_t jo = 0;
    for (int l = 0; l < 3; l++) {
        rc = launch_tree_jobs(s, d_jobs + jo, (u32)plan.jobs[l].size(), d_small);
        if (rc) return rc;
        jo += plan.jobs[l].size();
    }
    ECG_HIP_CHECK(hipMemcpyAsync(d_root, d_small + 32ull * plan.root_chunk, 32, hipMemcpyDeviceToDevice, s));
    c->last_hash64 = hc;
    return ECGPU_SUCCESS;
}

}  // namespace ecg

using namespace ecg;

extern "C" {

uint64_t ecgpu_beacon_state_deneb_fixed_size(int preset) {
    if (preset < 0 || preset > 1) return 0;
    return layout_for(STATE_PRESETS[preset]).size;
}

int ecgpu_htr_beacon_state_deneb_dev(const uint8_t* d_ssz, uint64_t n_bytes, const uint8_t* h_fixed, int preset,
                                     uint8_t* d_root, ecgpu_stream_t stream) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!d_ssz || !h_fixed || !d_root) return ECGPU_ERR_BAD_ARG;
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(stream);
    return state_root_device(s, c, d_ssz, n_bytes, h_fixed, preset, d_root);
}

int ecgpu_htr_beacon_state_deneb(const uint8_t* ssz, uint64_t n_bytes, int preset, uint8_t root[32]) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!ssz || !root) return ECGPU_ERR_BAD_ARG;
    if (preset < 0 || preset > 1 || n_bytes < layout_for(STATE_PRESETS[preset]).size) {
        set_last_error("bad preset or truncated state");
        return ECGPU_ERR_BAD_ARG;
    }
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    // the encoding lives in its own allocation (the arena is rebuilt by the driver)
    static thread_local u8* d_state = nullptr;
    static thread_local size_t d_state_cap = 0;
    if (n_bytes + 64 > d_state_cap) {
        if (d_state) {
            ECG_HIP_CHECK(hipStreamSynchronize(s));
            ECG_HIP_CHECK(hipFree(d_state));
            d_state = nullptr;
            d_state_cap = 0;
        }
        ECG_HIP_CHECK(hipMalloc((void**)&d_state, n_bytes + 64 + (n_bytes >> 3)));
        d_state_cap = n_bytes + 64 + (n_bytes >> 3);
    }
    ECG_HIP_CHECK(hipMemcpyAsync(d_state, ssz, n_bytes, hipMemcpyHostToDevice, s));
    u8* d_root = d_state + ((n_bytes + 31) / 32) * 32;
    rc = state_root_device(s, c, d_state, n_bytes, ssz, preset, d_root);
    if (rc) return rc;
    ECG_HIP_CHECK(hipMemcpyAsync(root, d_root, 32, hipMemcpyDeviceToHost, s));
    ECG_HIP_CHECK(hipStreamSynchronize(s));
    return ECGPU_SUCCESS;
}

}  // extern "C"
