// libecgpu.so runtime: see runtime.h.
#include "runtime.h"

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <thread>

namespace ecg {

static thread_local std::string t_last_error;
void set_last_error(const std::string& s) { t_last_error = s; }

static std::mutex g_init_mu;
static std::atomic<int> g_init_state{0};  // 0 = not tried, 1 = ok, -1 = no device
static int g_device = -1;                 // the process default: the device of the first ecgpu_init
// Several GPUs in ONE process (a Rust host has no torch.distributed): a host thread may bind itself to any device
// (ecgpu_bind_thread); device tables, streams and arenas exist per device, handles (registry, resident state, batch results)
// belong to the device their creating thread was bound to.
static thread_local int t_device = -1;
static std::atomic<int> g_dev_state[MAX_DEVICES];
int current_device() { return t_device >= 0 ? t_device : g_device; }

int init_merkle_tables(hipStream_t s);  // merkle.hip
int init_bls_tables(hipStream_t s);     // bls.hip

// tables of one device (zero-hash ladder, lane-group programs); g_init_mu held
static int init_device_locked(int device) {
    if (device < 0 || device >= MAX_DEVICES) return ECGPU_ERR_BAD_ARG;
    if (g_dev_state[device].load() == 1) return ECGPU_SUCCESS;
    ECG_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    ECG_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_last_error(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
        return ECGPU_ERR_NO_DEVICE;
    }
    const int saved = t_device;
    t_device = device;  // the table builders address "the current device"
    int rc = init_merkle_tables(nullptr);
    if (!rc) rc = init_bls_tables(nullptr);
    t_device = saved;
    if (rc) return rc;
    ECG_HIP_CHECK(hipDeviceSynchronize());
    g_dev_state[device].store(1);
    return ECGPU_SUCCESS;
}

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
    g_device = device;
    int rc = init_device_locked(device);
    if (rc) {
        if (rc == ECGPU_ERR_NO_DEVICE) g_init_state.store(-1);
        return rc;
    }
    g_init_state.store(1);
    return ECGPU_SUCCESS;
}

int ensure_init() {
    if (g_init_state.load() != 1) {
        int rc = do_init(-1);
        if (rc) return rc;
    }
    // bind this thread to its device (the process default unless ecgpu_bind_thread chose another)
    const int dev = current_device();
    if (g_dev_state[dev].load() != 1) {
        std::lock_guard<std::mutex> lk(g_init_mu);
        int rc = init_device_locked(dev);
        if (rc) return rc;
    }
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != dev) ECG_HIP_CHECK(hipSetDevice(dev));
    return ECGPU_SUCCESS;
}

// per (host thread, device): streams and arenas are device objects.  The holder's destructor runs when the thread exits
// (for the main thread: before static destructors, i.e. while the HIP runtime is still up) and gives everything back.
void ThreadCtx::release() {
    // The arenas are keyed by the stream they were used on, and some of those streams are the CALLER's: they may have been
    // destroyed long ago, so their handles are not touched here.  One device-wide synchronisation covers every stream that
    // could still be reading an arena (this runs once per thread, at its exit).
    bool any = own_stream != nullptr || aux.ready || !small_scratch.empty();
    for (auto& kv : arenas) any = any || kv.second.base != nullptr;
    if (any) (void)hipDeviceSynchronize();
    for (auto& kv : arenas)
        if (kv.second.base) (void)hipFree(kv.second.base);
    arenas.clear();
    for (auto& kv : small_scratch) (void)hipFree(kv.second);
    small_scratch.clear();
    if (staging.p) (void)hipHostFree(staging.p);
    staging = PinnedBuf();
    uploads.release();
    aux.give_back();
    if (own_stream) (void)hipStreamDestroy(own_stream);
    own_stream = nullptr;
}
namespace {
struct ThreadCtxs {
    std::map<int, ThreadCtx*> by_device;
    ~ThreadCtxs() {
        for (auto& kv : by_device) {
            if (hipSetDevice(kv.first) == hipSuccess) kv.second->release();
            delete kv.second;
        }
    }
};
}  // namespace
static thread_local ThreadCtxs t_ctxs;
ThreadCtx* tctx() {
    ThreadCtx*& c = t_ctxs.by_device[current_device()];
    if (!c) c = new ThreadCtx();
    return c;
}

// ---- persistent per-device workers (see runtime.h) ---------------------------------------------------------------------------
namespace {
struct DeviceWorker {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    std::thread th;
    void loop(int device) {
        (void)ecgpu_bind_thread(device);
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !q.empty(); });
                job = std::move(q.front());
                q.pop_front();
            }
            job();
        }
    }
};
std::mutex g_workers_mu;
DeviceWorker* g_workers[MAX_DEVICES];  // never destroyed: the threads sleep until the process ends
DeviceWorker* worker_of(int device) {
    std::lock_guard<std::mutex> lk(g_workers_mu);
    DeviceWorker*& w = g_workers[device];
    if (!w) {
        std::unique_ptr<DeviceWorker> fresh(new DeviceWorker());
        DeviceWorker* raw = fresh.get();
        fresh->th = std::thread([raw, device] { raw->loop(device); });  // may throw: the slot stays empty then
        fresh->th.detach();
        w = fresh.release();
    }
    return w;
}
}  // namespace

int run_on_devices(const int* devices, unsigned n, const std::function<int(unsigned)>& fn, std::vector<int>& rcs,
                   std::vector<std::string>& errs) {
    rcs.assign(n, 0);
    errs.assign(n, std::string());
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return ECGPU_ERR_NO_DEVICE;
    for (unsigned g = 0; g < n; g++)
        if (devices[g] < 0 || devices[g] >= n_dev || devices[g] >= MAX_DEVICES) {
            set_last_error("device index out of range");
            return ECGPU_ERR_BAD_ARG;
        }
    // Every worker exists before the first job is queued (creating one may throw: nothing may be in flight then), and the
    // completion state is owned jointly by the jobs and this frame, so a job never refers to a frame that has been unwound.
    std::vector<DeviceWorker*> workers(n, nullptr);
    try {
        for (unsigned g = 0; g < n; g++) workers[g] = worker_of(devices[g]);
    } catch (const std::exception& e) {
        set_last_error(std::string("could not start a device worker: ") + e.what());
        return ECGPU_ERR_HIP;
    }
    struct Completion {
        std::mutex mu;
        std::condition_variable cv;
        unsigned done = 0;
        std::vector<int> rcs;
        std::vector<std::string> errs;
    };
    auto st = std::make_shared<Completion>();
    st->rcs.assign(n, 0);
    st->errs.assign(n, std::string());
    const std::function<int(unsigned)>* pfn = &fn;  // outlives the wait below, which every job finishes before
    for (unsigned g = 0; g < n; g++) {
        DeviceWorker* w = workers[g];
        std::lock_guard<std::mutex> lk(w->mu);
        w->q.push_back([st, pfn, g] {
            int rc = ensure_init();  // the worker is bound to its device; this (re)selects it for the HIP runtime
            if (!rc) rc = (*pfn)(g);
            std::string err = rc ? ecgpu_last_error() : "";
            std::lock_guard<std::mutex> dl(st->mu);
            st->rcs[g] = rc;
            st->errs[g] = err;
            st->done++;
            st->cv.notify_one();
        });
        w->cv.notify_one();
    }
    std::unique_lock<std::mutex> lk(st->mu);
    st->cv.wait(lk, [&] { return st->done == n; });
    rcs = st->rcs;
    errs = st->errs;
    return ECGPU_SUCCESS;
}

hipStream_t ThreadCtx::stream_or_own(ecgpu_stream_t s) {
    if (s) return (hipStream_t)s;
    if (!own_stream) {
        if (hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking) != hipSuccess) own_stream = nullptr;
    }
    return own_stream;
}

// (never destroyed: a host thread may exit after the static destructors have run)
static std::mutex& g_aux_pool_mu = *new std::mutex();
static std::vector<AuxStreams>* const g_aux_pool = new std::vector<AuxStreams>[MAX_DEVICES];

static std::atomic<int> g_live_aux_sets{0};
int AuxStreams::live_sets() { return g_live_aux_sets.load(); }

void AuxStreams::give_back() {
    if (!ready) return;
    g_live_aux_sets.fetch_sub(1);
    {
        std::lock_guard<std::mutex> lk(g_aux_pool_mu);
        g_aux_pool[device].push_back(*this);
    }
    *this = AuxStreams();
}

int AuxStreams::init() {
    if (ready) return ECGPU_SUCCESS;
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> lk(g_aux_pool_mu);
        auto& pool = g_aux_pool[dev];
        if (!pool.empty()) {
            *this = pool.back();
            pool.pop_back();
            g_live_aux_sets.fetch_add(1);
            return ECGPU_SUCCESS;
        }
    }
    // a half-built set is released, not kept (advisor, round 3: the same rule as for the upload ring)
    auto build = [this]() -> int {
        ECG_HIP_CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        for (int i = 0; i < N_AUX_STREAMS; i++) {
            // st[AUX_SIG] sits at the high priority level: the runtime multiplexes normal-priority streams onto four hardware
            // queues and st[1] / st[2] ended up sharing one (a signature stage on st[1] ran AFTER the message stage on st[2],
            // not beside it); each priority level has queues of its own.  ECGPU_AUX1_PRIORITY=0: all at normal priority.
            static const int aux1_high = [] { const char* e = getenv("ECGPU_AUX1_PRIORITY"); return e ? atoi(e) : 1; }();
            if ((i == AUX_SIG && aux1_high) || (i == 2 && aux1_high >= 2)) {
                int lo = 0, hi = 0;
                ECG_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
                ECG_HIP_CHECK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, hi));
            } else {
                ECG_HIP_CHECK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
            }
            ECG_HIP_CHECK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
            ECG_HIP_CHECK(hipEventCreateWithFlags(&reached[i], hipEventDisableTiming));
        }
        return ECGPU_SUCCESS;
    };
    device = dev;
    const int rc = build();
    if (rc) {
        if (fork) (void)hipEventDestroy(fork);
        for (int i = 0; i < N_AUX_STREAMS; i++) {
            if (done[i]) (void)hipEventDestroy(done[i]);
            if (reached[i]) (void)hipEventDestroy(reached[i]);
            if (st[i]) (void)hipStreamDestroy(st[i]);
        }
        *this = AuxStreams();
        return rc;
    }
    ready = true;
    g_live_aux_sets.fetch_add(1);
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

int UploadRing::acquire(size_t bytes, u8** slot, hipEvent_t* ev) {
    if (bytes > slot_bytes) {
        release();
        const size_t want = (bytes + 4095) & ~(size_t)4095;
        ECG_HIP_CHECK(hipHostMalloc((void**)&p, want * UPLOAD_SLOTS, hipHostMallocDefault));
        for (int i = 0; i < UPLOAD_SLOTS; i++) copied[i] = nullptr;
        for (int i = 0; i < UPLOAD_SLOTS; i++)
            if (hipEventCreateWithFlags(&copied[i], hipEventDisableTiming) != hipSuccess) {
                copied[i] = nullptr;
                release();  // a half-built ring would fail every later call at hipEventSynchronize(nullptr)
                set_last_error("hipEventCreate failed while building the upload ring");
                return ECGPU_ERR_HIP;
            }
        slot_bytes = want;
    } else {
        ECG_HIP_CHECK(hipEventSynchronize(copied[next]));  // never recorded: returns at once
    }
    *slot = p + slot_bytes * next;
    *ev = copied[next];
    next = (next + 1) % UPLOAD_SLOTS;
    return ECGPU_SUCCESS;
}
void UploadRing::release() {
    if (!p) return;
    for (int i = 0; i < UPLOAD_SLOTS; i++) {
        if (!copied[i]) continue;
        (void)hipEventSynchronize(copied[i]);
        (void)hipEventDestroy(copied[i]);
        copied[i] = nullptr;
    }
    (void)hipHostFree(p);
    p = nullptr;
    slot_bytes = 0;
    next = 0;
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

int ecgpu_bind_thread(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ECGPU_ERR_NO_DEVICE;
    if (device < 0 || device >= n || device >= MAX_DEVICES) return ECGPU_ERR_BAD_ARG;
    if (g_init_state.load() != 1) {
        int rc = do_init(device);
        if (rc) return rc;
    }
    t_device = device;
    return ensure_init();
}

int ecgpu_thread_device(void) {
    if (g_init_state.load() != 1) return ECGPU_ERR_NO_DEVICE;
    return current_device();
}

int ecgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* ecgpu_version(void) { return "ecgpu 0.4 (gfx950)"; }

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
