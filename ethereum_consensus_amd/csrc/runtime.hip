// libecgpu.so runtime: see runtime.h.
#include "runtime.h"

#include <atomic>
#include <cstring>

namespace ecg {

static thread_local std::string t_last_error;
void set_last_error(const std::string& s) { t_last_error = s; }

static std::mutex g_init_mu;
static std::atomic<int> g_init_state{0};  // 0 = not tried, 1 = ok, -1 = no device
static int g_device = -1;

int init_merkle_tables(hipStream_t s);  // merkle.hip
int init_bls_tables(hipStream_t s);     // bls.hip

static int do_init(int device) {
    std::lock_guard<std::mutex> lk(g_init_mu);
    if (g_init_state.load() == 1) return ECGPU_SUCCESS;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_last_error("no HIP device visible: libecgpu has no CPU fallback");
        g_init_state.store(-1);
        return ECGPU_ERR_NO_DEVICE;
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= n) {
        set_last_error("device index out of range");
        return ECGPU_ERR_BAD_ARG;
    }
    ECG_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    ECG_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_last_error(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
        g_init_state.store(-1);
        return ECGPU_ERR_NO_DEVICE;
    }
    g_device = device;
    int rc = init_merkle_tables(nullptr);
    if (rc) return rc;
    rc = init_bls_tables(nullptr);
    if (rc) return rc;
    ECG_HIP_CHECK(hipDeviceSynchronize());
    g_init_state.store(1);
    return ECGPU_SUCCESS;
}

int ensure_init() {
    int st = g_init_state.load();
    if (st == 1) {
        // bind this thread to the library's device
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != g_device) (void)hipSetDevice(g_device);
        return ECGPU_SUCCESS;
    }
    return do_init(-1);
}

static thread_local ThreadCtx* t_ctx = nullptr;
ThreadCtx* tctx() {
    if (!t_ctx) t_ctx = new ThreadCtx();  // lives for the thread; a handful of bytes + arenas
    return t_ctx;
}

hipStream_t ThreadCtx::stream_or_own(ecgpu_stream_t s) {
    if (s) return (hipStream_t)s;
    if (!own_stream) {
        if (hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking) != hipSuccess) own_stream = nullptr;
    }
    return own_stream;
}

int AuxStreams::init() {
    if (ready) return ECGPU_SUCCESS;
    ECG_HIP_CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (int i = 0; i < N_AUX_STREAMS; i++) {
        ECG_HIP_CHECK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
        ECG_HIP_CHECK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    }
    ready = true;
    return ECGPU_SUCCESS;
}

int Arena::reserve(size_t bytes) {
    if (bytes <= cap) return ECGPU_SUCCESS;
    if (base) {
        ECG_HIP_CHECK(hipDeviceSynchronize());
        ECG_HIP_CHECK(hipFree(base));
        base = nullptr;
        cap = 0;
    }
    size_t want = bytes + (bytes >> 2) + (1u << 20);
    ECG_HIP_CHECK(hipMalloc((void**)&base, want));
    cap = want;
    used = 0;
    return ECGPU_SUCCESS;
}

u8* Arena::take(size_t bytes, size_t align) {
    size_t off = (used + align - 1) / align * align;
    if (off + bytes > cap) return nullptr;
    used = off + bytes;
    return base + off;
}

int PinnedBuf::reserve(size_t bytes) {
    if (bytes <= cap) return ECGPU_SUCCESS;
    if (p) {
        ECG_HIP_CHECK(hipHostFree(p));
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes + (bytes >> 2) + 4096;
    ECG_HIP_CHECK(hipHostMalloc((void**)&p, want, hipHostMallocDefault));
    cap = want;
    return ECGPU_SUCCESS;
}

// ---- profiling ------------------------------------------------------------------------------
struct ProfRec {
    std::string tag;
    hipEvent_t a, b;
};
static std::mutex g_prof_mu;
static std::atomic<int> g_prof_on{0};
static std::vector<ProfRec> g_prof;
static std::string g_prof_filter;  // empty = every tagged kernel

ProfScope::ProfScope(const char* t, hipStream_t st) : tag(t), s(st), on(g_prof_on.load() != 0) {
    if (!on) return;
    if (!g_prof_filter.empty() && g_prof_filter != t) {
        on = false;
        return;
    }
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        on = false;
        return;
    }
    (void)hipEventRecord(a, s);
}
ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(b, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back({tag, a, b});
}

}  // namespace ecg

using namespace ecg;

extern "C" {

int ecgpu_init(int device) {
    if (g_init_state.load() == 1) return ensure_init();
    return do_init(device);
}

int ecgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* ecgpu_version(void) { return "ecgpu 0.1 (gfx950)"; }

const char* ecgpu_last_error(void) { return t_last_error.c_str(); }

int ecgpu_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    g_prof_on.store(on);
    return ECGPU_SUCCESS;
}

int ecgpu_prof_filter(const char* kernel_tag) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_filter = kernel_tag ? kernel_tag : "";
    return ECGPU_SUCCESS;
}

int ecgpu_prof_read(const char* kernel_tag, double* total_ms, uint64_t* launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double tot = 0;
    uint64_t n = 0;
    for (auto& r : g_prof) {
        if (kernel_tag && r.tag != kernel_tag) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            tot += ms;
            n++;
        }
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return (int)n;
}

}  // extern "C"
