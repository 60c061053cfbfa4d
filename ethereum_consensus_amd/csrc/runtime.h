// Host-side runtime of libecgpu.so: device binding, per-thread stream + workspace arenas, error
// reporting and HIP-event kernel timing.  There is deliberately no CPU execution path here.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ecgpu.h"
#include "common.h"

namespace ecg {

void set_last_error(const std::string& s);

#define ECG_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            ecg::set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e));          \
            return _e == hipErrorOutOfMemory ? ECGPU_ERR_OOM                                 \
                   : (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? ECGPU_ERR_NO_DEVICE \
                                                                              : ECGPU_ERR_HIP; \
        }                                                                                    \
    } while (0)

// A growable device arena owned by one (thread, stream) pair.  Offsets are handed out bump-style
// per call; nothing is freed until the thread exits, so steady-state calls do no hipMalloc.
struct Arena {
    u8* base = nullptr;
    size_t cap = 0;
    size_t used = 0;
    int reserve(size_t bytes);            // ensure capacity (may reallocate: only legal when used == 0)
    u8* take(size_t bytes, size_t align = 256);
    void reset() { used = 0; }
};

struct PinnedBuf {
    u8* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
};

// Fork/join helper: a few auxiliary streams per host thread so that independent launch chains of one
// call (the fields of a BeaconState) overlap on the device instead of queueing behind each other.
constexpr int N_AUX_STREAMS = 4;
constexpr int AUX_SIG = 3;  // the signature stage's stream: high priority = a hardware queue of its own (runtime.hip)
struct AuxStreams {
    hipStream_t st[N_AUX_STREAMS] = {};
    hipEvent_t fork = nullptr, done[N_AUX_STREAMS] = {}, reached[N_AUX_STREAMS] = {};  // reached[i]: st[i] has passed its wait on fork
    bool ready = false;
    int device = -1;
    int init();     // takes a set from the device's pool of sets that exited threads gave back, or creates one
    static int live_sets();  // stream sets currently held by host threads of this process (all devices)
    void give_back();  // (thread exit, after a device-wide synchronisation) back to the pool: streams are never destroyed --
                       // short-lived host threads would otherwise create and destroy hardware queues at every exit
};

// A ring of pinned host slots for small descriptor blocks that travel to the device in front of a launch chain (the plan of a
// BeaconState root): the copy is truly asynchronous (pageable memory would go through the runtime's staging path), and a
// slot is rewritten only after the copy that read it has completed (an event per slot; a host running more than
// UPLOAD_SLOTS calls ahead of the device waits here).
constexpr int UPLOAD_SLOTS = 8;
struct UploadRing {
    u8* p = nullptr;
    size_t slot_bytes = 0;
    hipEvent_t copied[UPLOAD_SLOTS] = {};
    int next = 0;
    int acquire(size_t bytes, u8** slot, hipEvent_t* ev);  // the caller fills *slot, enqueues the copy and records *ev after it
    void release();
};

struct ThreadCtx {
    hipStream_t own_stream = nullptr;
    // ONE set of auxiliary streams per host thread (three at normal priority + one at high priority).  BLS batches run their
    // message stage on st[2] and their signature stage on st[AUX_SIG] (rounds 1-3: st[1], which turned out to share a hardware
    // queue with st[2]); the state root has needed none since its tail became one kernel (round 3).  Not a set per purpose:
    // the runtime multiplexes normal-priority streams onto 4 hardware queues, and with more live streams than that the
    // auxiliary ones start sharing a queue with the caller's stream (measured in round 1: profiles/r01s16_hw_queues.txt).
    AuxStreams aux;
    std::map<hipStream_t, Arena> arenas;
    PinnedBuf staging;      // host-pinned staging for small H2D/D2H payloads
    UploadRing uploads;
    u64 last_hash64 = 0;
    // 4 KB of device memory outside the arenas PER STREAM (a block that must outlive an arena reset; advisor, round 4: one
    // buffer per thread was shared by asynchronous calls of that thread on different streams)
    std::map<hipStream_t, u8*> small_scratch;
    hipStream_t stream_or_own(ecgpu_stream_t s);
    Arena& arena(hipStream_t s) { return arenas[s]; }
    void release();  // synchronize and free every stream, event, arena and pinned buffer (thread exit)
};

constexpr int MAX_DEVICES = 16;
int current_device();       // the device the calling thread is bound to (ecgpu_bind_thread; default: the process's first)
int ensure_init();          // ECGPU_SUCCESS or ECGPU_ERR_NO_DEVICE; binds the thread to current_device()
ThreadCtx* tctx();          // per-thread context (after ensure_init); freed when the thread exits

// One persistent worker thread per device for the *_multi entries (several GPUs under one host process): created on first
// use, bound to its device once, reused by every later call -- so its streams, arenas and pinned staging are too.  (Round 2
// started fresh std::threads per call, each of which built a context that nothing released: a host calling a *_multi entry
// once per slot leaked streams and HBM on every device.)  run_on_devices runs fn(g) for g in [0, n) on devices[g]'s worker
// and returns when all are done; entries for the same device run one after the other.
int run_on_devices(const int* devices, unsigned n, const std::function<int(unsigned)>& fn, std::vector<int>& rcs,
                   std::vector<std::string>& errs);

// kernel timing with HIP events on the launch stream
struct ProfScope {
    ProfScope(const char* tag, hipStream_t s);
    ~ProfScope();
    const char* tag;
    hipStream_t s;
    hipEvent_t a = nullptr, b = nullptr;
    bool on;
};

}  // namespace ecg
