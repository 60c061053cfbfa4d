// tests/hostsim: csrc/state_fields.h -- the field-addressed write queue of a resident BeaconState -- over a HOST byte array.
// TEST INFRASTRUCTURE ONLY.  The product runs the same FieldWriter over the device-resident encoding (state_deneb.hip
// ResidentSink); here the sink is a std::vector, so that the (field, index) -> byte arithmetic, the queue's program-order
// semantics and the later-write-wins resolution can be executed and compared with the oracle without a GPU.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "state_fields.h"

using namespace ecg;

namespace {
struct HostState {
    int fork_, preset_;
    std::vector<u8> enc;
    std::string error;
    u32 patch_calls = 0, resize_calls = 0, patches = 0;
    FieldWriter<HostState> queue;

    int fork() const { return fork_; }
    int preset() const { return preset_; }
    const u8* fixed() const { return enc.data(); }
    u64 size() const { return enc.size(); }
    void fail(const char* m) { error = m; }

    int apply_patches(const u64* offsets, const u64* data_off, const u8* data, u32 n) {
        patch_calls++;
        patches += n;
        // the contract of ecgpu_resident_state_patch: inside the encoding, pairwise disjoint (they are applied concurrently)
        std::vector<std::pair<u64, u64>> r;
        for (u32 i = 0; i < n; i++) {
            const u64 len = data_off[i + 1] - data_off[i];
            if (offsets[i] > enc.size() || len > enc.size() - offsets[i]) return fail("patch outside the encoding"), -3;
            r.push_back({offsets[i], offsets[i] + len});
        }
        std::sort(r.begin(), r.end());
        for (size_t i = 1; i < r.size(); i++)
            if (r[i].first < r[i - 1].second) return fail("overlapping patches in one call"), -3;
        for (u32 i = 0; i < n; i++) std::memcpy(enc.data() + offsets[i], data + data_off[i], data_off[i + 1] - data_off[i]);
        return 0;
    }
    int apply_resize(u32 vi, const u8* data, u64 add_len, u64 keep, FieldResize mode) {
        resize_calls++;
        u64 words[N_STATE_VAR_FIELDS];
        state_offset_words(fork_, preset_, words);
        if (vi >= (u32)N_STATE_VAR_FIELDS || words[vi] == NO_FIELD) return fail("no such list"), -3;
        const u64 start = rd32(enc.data() + words[vi]);
        u64 end = enc.size();
        for (int k = (int)vi + 1; k < N_STATE_VAR_FIELDS; k++)
            if (words[k] != NO_FIELD) {
                end = rd32(enc.data() + words[k]);
                break;
            }
        std::vector<u8> next(enc.begin(), enc.begin() + start);
        if (mode == FIELD_APPEND) {
            next.insert(next.end(), enc.begin() + start, enc.begin() + end);
            next.insert(next.end(), data, data + add_len);
        } else if (mode == FIELD_TRUNCATE) {
            if (keep > end - start) return fail("truncate beyond the list"), -3;
            next.insert(next.end(), enc.begin() + start, enc.begin() + start + keep);
        } else {
            next.insert(next.end(), data, data + add_len);
        }
        const int64_t delta = (int64_t)(next.size() - start) - (int64_t)(end - start);
        next.insert(next.end(), enc.begin() + end, enc.end());
        for (int k = (int)vi + 1; k < N_STATE_VAR_FIELDS; k++)
            if (words[k] != NO_FIELD) {
                const u32 v = (u32)((int64_t)rd32(next.data() + words[k]) + delta);
                for (int b = 0; b < 4; b++) next[words[k] + b] = (u8)(v >> (8 * b));
            }
        // like the product: the resized state must still be a state of this fork, else nothing changes
        StatePlan probe;
        u8 ext[64] = {};
        if (!build_state_plan(fork_, next.data(), next.size(), preset_, probe, fork_ == FORK_PHASE0 ? ext : nullptr, nullptr)) return fail(probe.error.c_str()), -3;
        enc.swap(next);
        return 0;
    }
    int apply_rotate(u64 prev_start, u64 cur_start, u64 len) {
        std::memmove(enc.data() + prev_start, enc.data() + cur_start, len);
        std::memset(enc.data() + cur_start, 0, len);
        return 0;
    }
};
}  // namespace

extern "C" {

void* hs_fs_create(int fork, int preset, const u8* ssz, u64 n) {
    HostState* h = new HostState();
    h->fork_ = fork, h->preset_ = preset;
    h->enc.assign(ssz, ssz + n);
    return h;
}
void hs_fs_destroy(void* p) { delete (HostState*)p; }
int hs_fs_patch_field(void* p, u32 field, u64 off, const u8* data, u64 n) {
    HostState* h = (HostState*)p;
    return h->queue.write(*h, field, off, data, n);
}
int hs_fs_patch_elements(void* p, u32 field, u64 first, const u8* data, u64 n) {
    HostState* h = (HostState*)p;
    const FieldStatic f = field_static(h->fork_, h->preset_, field);
    if (!f.present || !f.elem || n % f.elem) return -3;
    return h->queue.write(*h, field, first * f.elem, data, n);
}
int hs_fs_push(void* p, u32 field, const u8* data, u64 n) {
    HostState* h = (HostState*)p;
    return h->queue.push(*h, field, data, n);
}
int hs_fs_truncate_field(void* p, u32 field, u64 keep) {
    HostState* h = (HostState*)p;
    return h->queue.truncate(*h, field, keep);
}
int hs_fs_set_field(void* p, u32 field, const u8* data, u64 n) {
    HostState* h = (HostState*)p;
    return h->queue.set(*h, field, data, n);
}
int hs_fs_add_validator(void* p, const u8* rec, u64 balance) {
    HostState* h = (HostState*)p;
    return h->queue.add_validator(*h, rec, balance);
}
int hs_fs_rotate_participation(void* p) {
    HostState* h = (HostState*)p;
    return h->queue.rotate_participation(*h);
}
int hs_fs_flush(void* p) {
    HostState* h = (HostState*)p;
    return h->queue.flush(*h);
}
long long hs_fs_field_size(void* p, u32 field) {
    HostState* h = (HostState*)p;
    FieldLoc loc;
    u64 seen = 0;
    if (!h->queue.locate(*h, field, loc, seen)) return -3;
    return (long long)seen;
}
// the encoding as applied so far (call hs_fs_flush first for the program-order view); returns its size
u64 hs_fs_encoding(void* p, u8* out, u64 cap) {
    HostState* h = (HostState*)p;
    if (out && cap >= h->enc.size()) std::memcpy(out, h->enc.data(), h->enc.size());
    return h->enc.size();
}
// how the queue reached the sink: [0] patch calls, [1] resize calls, [2] patches handed over
void hs_fs_counters(void* p, u32 out[3]) {
    HostState* h = (HostState*)p;
    out[0] = h->patch_calls, out[1] = h->resize_calls, out[2] = h->patches;
}
const char* hs_fs_error(void* p) { return ((HostState*)p)->error.c_str(); }

}  // extern "C"
