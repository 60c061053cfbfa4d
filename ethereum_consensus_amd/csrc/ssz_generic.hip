// hash_tree_root of any SSZ value from its serialization and a type description (include/ecgpu.h
// ecgpu_htr_ssz; plan: ssz_plan.h).  The host walks SSZ offsets and emits descriptors; every hash64 runs on the
// GPU: basic values and packed bytes are gathered into one chunk buffer, each dependency level of containers /
// lists is ONE batched k_tree_jobs launch, sequences above 512 chunks go through the pass / tile kernels.
#include <vector>

#include "merkle_driver.h"
#include "ssz_plan.h"

using namespace ecg;

namespace ecg {
int htr_ssz_impl(const ecgpu_ssz_type* types, uint32_t n_types, const uint32_t* fields, uint32_t n_field_refs, uint32_t root_type,
                 const uint8_t* ssz, uint64_t n_bytes, uint8_t root[32], bool allow_internal);
}
extern "C" int ecgpu_htr_ssz(const ecgpu_ssz_type* types, uint32_t n_types, const uint32_t* fields, uint32_t n_field_refs,
                             uint32_t root_type, const uint8_t* ssz, uint64_t n_bytes, uint8_t root[32]) {
    return ecg::htr_ssz_impl(types, n_types, fields, n_field_refs, root_type, ssz, n_bytes, root, false);
}
// allow_internal: the type table may contain ECG_SSZ_LIST_NOMIX (ssz_proof.hip)
int ecg::htr_ssz_impl(const ecgpu_ssz_type* types, uint32_t n_types, const uint32_t* fields, uint32_t n_field_refs, uint32_t root_type,
                      const uint8_t* ssz, uint64_t n_bytes, uint8_t root[32], bool allow_internal) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!types || !root || (!ssz && n_bytes)) return ECGPU_ERR_BAD_ARG;
    SszPlan plan;
    static const u8 empty[4] = {0, 0, 0, 0};
    if (!build_ssz_plan(types, n_types, fields, n_field_refs, root_type, ssz ? ssz : empty, n_bytes, plan, allow_internal)) {
        set_last_error(plan.error);
        return ECGPU_ERR_BAD_ARG;
    }
    ThreadCtx* c = tctx();
    hipStream_t s = c->stream_or_own(nullptr);
    Arena& ar = c->arena(s);
    ar.reset();
    size_t n_jobs = 0;
    for (auto& l : plan.jobs) n_jobs += l.size();
    const size_t small_bytes = 32ull * plan.n_chunks;
    size_t need = n_bytes + 64 + small_bytes + n_jobs * sizeof(TreeJob) + plan.gathers.size() * sizeof(GatherDesc) + 8192;
    for (auto& b : plan.bigs) need += merkle_ws_bytes(b.n0) + 512;
    rc = ar.reserve(need);
    if (rc) return rc;
    u8* d_ssz = ar.take(n_bytes + 4);
    u8* d_small = ar.take(small_bytes);
    TreeJob* d_jobs = (TreeJob*)ar.take((n_jobs ? n_jobs : 1) * sizeof(TreeJob));
    GatherDesc* d_gath = (GatherDesc*)ar.take((plan.gathers.size() ? plan.gathers.size() : 1) * sizeof(GatherDesc));
    if (!d_ssz || !d_small || !d_jobs || !d_gath) return ECGPU_ERR_OOM;
    if (n_bytes) ECG_HIP_CHECK(hipMemcpyAsync(d_ssz, ssz, n_bytes, hipMemcpyHostToDevice, s));
    std::vector<TreeJob> all_jobs;
    std::vector<size_t> level_start(plan.jobs.size() + 1, 0);
    for (size_t l = 0; l < plan.jobs.size(); l++) {
        level_start[l] = all_jobs.size();
        all_jobs.insert(all_jobs.end(), plan.jobs[l].begin(), plan.jobs[l].end());
    }
    level_start[plan.jobs.size()] = all_jobs.size();
    if (n_jobs) ECG_HIP_CHECK(hipMemcpyAsync(d_jobs, all_jobs.data(), n_jobs * sizeof(TreeJob), hipMemcpyHostToDevice, s));
    if (!plan.gathers.empty())
        ECG_HIP_CHECK(hipMemcpyAsync(d_gath, plan.gathers.data(), plan.gathers.size() * sizeof(GatherDesc), hipMemcpyHostToDevice, s));
    ECG_HIP_CHECK(hipMemsetAsync(d_small, 0, small_bytes, s));
    rc = launch_gather(s, d_ssz, n_bytes, d_gath, (u32)plan.gathers.size(), d_small);
    if (rc) return rc;
    u64 hc = 0;
    u32 max_level = (u32)plan.jobs.size();
    for (auto& b : plan.bigs)
        if (b.level + 1 > max_level) max_level = b.level + 1;
    for (u32 l = 1; l < max_level; l++) {
        for (auto& b : plan.bigs) {
            if (b.level != l) continue;
            u8* ws = ar.take(merkle_ws_bytes(b.n0));
            if (!ws) return ECGPU_ERR_OOM;
            const u8* src = b.kind == LEAF_NODES ? d_small + 32ull * b.src : d_ssz + b.src;
            rc = merkleize_device(s, b.kind, src, b.bytes, b.n0, b.depth, b.mix, b.mix_len, d_small + 32ull * b.out_chunk, ws, &hc);
            if (rc) return rc;
        }
        if (l < plan.jobs.size()) {
            rc = launch_tree_jobs(s, d_jobs + level_start[l], (u32)plan.jobs[l].size(), d_small);
            if (rc) return rc;
        }
    }
    c->last_hash64 = plan.hashes;
    ECG_HIP_CHECK(hipMemcpyAsync(root, d_small, 32, hipMemcpyDeviceToHost, s));
    ECG_HIP_CHECK(hipStreamSynchronize(s));
    return ECGPU_SUCCESS;
}
