// capi.cpp -- extern "C" boundary of libtroyhip.so (include/troyhip.h).  No exceptions cross it.
#include <sys/random.h>
#include <cerrno>
#include "../../include/troyhip.h"
#include "evaluator.h"
#include "kernels.h"
#include "build_id.h"
#include "hostcrypto.h"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <mutex>
#include <cstring>
#include <string>

using namespace troyhip;

struct troyhip_context {
    Context ctx;
    Evaluator ev;
    troyhip_context(int scheme, u64 N, const std::vector<u64> &q, u64 t, bool device = true) : ctx(scheme, N, q, t, device), ev(ctx) {}
};

namespace {
thread_local std::string g_err;
std::atomic<bool> g_init{false};

template <class F> int guard(F f, bool need_init = true) {
    try {
        if (need_init && !g_init.load()) throw Error(ST_NOT_INITIALIZED, "KernelProvider not initialized.");
        f();
        // TROYHIP_SYNC=1 (the fifth documented switch): every entry point returns with the device idle -- what CUDA_LAUNCH_BLOCKING is to the reference.
        // For callers that read host timers around single calls without synchronising (the reference's own test/timetest.cu does) and for debugging;
        // the default is asynchronous, stream-ordered execution.
        static const bool sync_calls = [] { const char *e = std::getenv("TROYHIP_SYNC"); return e && e[0] == '1'; }();
        if (sync_calls && need_init && g_init.load()) HIP_CHECK(hipDeviceSynchronize());
        return TROYHIP_OK;
    } catch (const Error &e) {
        g_err = e.what();
        return e.code;
    } catch (const std::bad_alloc &) {
        g_err = "out of host memory";
        return TROYHIP_RUNTIME_ERROR;
    } catch (const std::exception &e) {
        g_err = e.what();
        return TROYHIP_RUNTIME_ERROR;
    }
}
// More than one GPU in one process (round 6): a context belongs to the device that was current when it was created, and every entry point that takes a
// context makes that device the calling thread's current one before anything else runs (HIP's current device is per thread: pool, streams, launches and
// device_cus() all follow it) -- so a host thread per device, or one thread walking over several contexts, both work; the thread is LEFT bound to it.
inline void bind_device(int device) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
    if (cur != device) HIP_CHECK(hipSetDevice(device));
}
// a null context is an argument error like any other (every use below sits inside guard(): the caller gets INVALID_ARGUMENT, not a fault)
template <class C> C *need(C *ctx) {
    if (!ctx) throw Error(ST_INVALID_ARGUMENT, "null context");
    if (ctx->ctx.has_device && g_init.load()) bind_device(ctx->ctx.device);
    return ctx;
}
CtBatch view(const troyhip_ct *c) {
    if (!c) throw Error(ST_INVALID_ARGUMENT, "null ciphertext descriptor");
    CtBatch b;
    b.data = c->data; b.bstride = c->batch_stride; b.size = c->size; b.limbs = c->limbs;
    b.ntt = c->is_ntt_form != 0; b.scale = c->scale; b.cf = c->correction_factor;
    return b;
}
void store(const CtBatch &b, troyhip_ct *c) {
    c->size = b.size; c->limbs = b.limbs; c->is_ntt_form = b.ntt ? 1 : 0; c->scale = b.scale; c->correction_factor = b.cf;
}
LimbMap map_from_primes(const Context &c, const u64 *row_primes, int period, int inner) {
    if (!row_primes || period < 1 || period > 64 || inner < 1) throw Error(ST_INVALID_ARGUMENT, "invalid limb description");
    std::vector<uint8_t> ids;
    for (int i = 0; i < period; i++) ids.push_back((uint8_t)c.prime_id(row_primes[i]));
    return c.ids_map(ids, (uint32_t)inner);
}
struct Timer { hipEvent_t a, b; };
} // namespace

#ifndef TROYHIP_CPU_EMUL
namespace troyhip { namespace ktime {
std::atomic<bool> enabled{false};
thread_local const char *tag = nullptr;
namespace {
struct Rec { std::string name; hipEvent_t a, b; };
std::vector<Rec> recs;
std::mutex kmu;
thread_local hipEvent_t open_stop = nullptr; // the stop event of the record THIS thread opened (two host threads launching concurrently
                                             // would otherwise attach their stop to each other's record)
}
void begin(const char *name, hipStream_t s) {
    Rec r;
    r.name = tag ? tag : name;
    tag = nullptr;
    HIP_CHECK(hipEventCreate(&r.a));
    HIP_CHECK(hipEventCreate(&r.b));
    HIP_CHECK(hipEventRecord(r.a, s));
    open_stop = r.b;
    std::lock_guard<std::mutex> g(kmu);
    recs.push_back(r);
}
void end(hipStream_t s) {
    if (open_stop) HIP_CHECK(hipEventRecord(open_stop, s));
    open_stop = nullptr;
}
static std::string report() {
    std::lock_guard<std::mutex> g(kmu);
    std::map<std::string, std::pair<unsigned, double>> agg; // name -> (calls, total microseconds)
    std::vector<std::string> order;
    for (Rec &r : recs) {
        float ms = 0;
        HIP_CHECK(hipEventSynchronize(r.b));
        HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
        if (!agg.count(r.name)) order.push_back(r.name);
        agg[r.name].first++;
        agg[r.name].second += ms * 1e3;
    }
    recs.clear();
    std::string out = "[";
    for (size_t i = 0; i < order.size(); i++) {
        char buf[96];
        std::snprintf(buf, sizeof(buf), "\", \"calls\": %u, \"total_us\": %.3f}", agg[order[i]].first, agg[order[i]].second);
        std::string nm;
        for (char ch : order[i]) if (ch != '"' && ch != '\\') nm += ch;
        out += (i ? ", {\"name\": \"" : "{\"name\": \"") + nm + buf;
    }
    return out + "]";
}
} }
#endif

extern "C" {

static std::atomic<uint64_t> g_stream_epoch{0}; // bumped whenever the pool forgets a stream: entry points then announce theirs again (on())
int troyhip_initialize(int device) {
    return guard([&] {
        HIP_CHECK(hipSetDevice(device));
        g_init.store(true);
    }, false);
}
int troyhip_is_initialized(void) { return g_init.load() ? 1 : 0; }
int troyhip_device_count(int *count) {
    return guard([&] { if (!count) throw Error(ST_INVALID_ARGUMENT, "null"); HIP_CHECK(hipGetDeviceCount(count)); }, false);
}
int troyhip_set_device(int device) { return guard([&] { HIP_CHECK(hipSetDevice(device)); }); }
int troyhip_get_device(int *device) { return guard([&] { if (!device) throw Error(ST_INVALID_ARGUMENT, "null"); HIP_CHECK(hipGetDevice(device)); }); }
const char *troyhip_last_error(void) { return g_err.c_str(); }
const char *troyhip_build_info(void) {
#ifdef TROYHIP_CPU_EMUL
    return "cpu-emulation (test only)";
#else
    return "gfx950";
#endif
}
const char *troyhip_build_id(void) { return TROYHIP_BUILD_ID; }
// KernelProvider::malloc / free behind the reference's MemoryPoolCuda policy (src/utils/memorypool_cuda.cuh:40-58): a freed
// block is kept and handed out again to a request of size <= block <= 2 * size; everything cached is released when the device
// runs short.  hipMalloc / hipFree synchronise the device, a pooled pair does not.  troyhip_free carries no stream, so reuse is
// ordered against every stream the pool KNOWS: the default stream, the streams created through troyhip_stream_create and the
// streams announced with troyhip_stream_register (a caller's own hipStream_t, a torch / RCCL stream).  A freed block is stamped
// with an event on each of them and is handed out again only once all have passed, so a block dropped by the host while a kernel
// on a known stream still reads it is never overwritten (the reference's pool leaves this to the caller).  Work on a stream the
// pool was never told about is NOT covered: register it, or synchronise it before freeing what it uses.  Thread-safe.
namespace {
struct DevicePool {
    struct Block { void *p; std::vector<std::pair<hipStream_t, hipEvent_t>> pending; }; // (stream, its free-point event)
    std::mutex mu;
    std::multimap<size_t, Block> free_blocks;
    std::map<void *, size_t> live;
    std::vector<hipStream_t> streams;   // the non-default streams the pool orders reuse against (troyhip_stream_create / _register)
    std::vector<hipEvent_t> spare_events;
    hipEvent_t new_event() {
        if (!spare_events.empty()) { hipEvent_t e = spare_events.back(); spare_events.pop_back(); return e; }
        hipEvent_t e;
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return e;
    }
    bool quiescent(Block &b) { // every tracked stream has passed the point at which the block was freed
        for (size_t i = b.pending.size(); i-- > 0;) {
            if (hipEventQuery(b.pending[i].second) != hipSuccess) { (void)hipGetLastError(); continue; }
            spare_events.push_back(b.pending[i].second);
            b.pending.erase(b.pending.begin() + (long)i);
        }
        return b.pending.empty();
    }
    void *get(size_t bytes) {
        std::lock_guard<std::mutex> g(mu);
        for (auto it = free_blocks.lower_bound(bytes); it != free_blocks.end() && it->first <= 2 * bytes; ++it) {
            if (!quiescent(it->second)) continue;
            void *p = it->second.p;
            live[p] = it->first;
            free_blocks.erase(it);
            return p;
        }
#ifndef TROYHIP_CPU_EMUL
        // No block of this size has been passed by every stream yet (the host runs ahead of the device: a loop that allocates a result per
        // step): rather than allocating again -- hipMalloc of a few hundred MB costs tens of milliseconds, and the cache would grow by one
        // block per step -- take a pending block and make every stream wait for the point at which it was freed (stream-ordered reuse).
        for (auto it = free_blocks.lower_bound(bytes); it != free_blocks.end() && it->first <= 2 * bytes; ++it) {
            Block &b = it->second;
            for (auto &se : b.pending) {
                for (size_t i = 0; i <= streams.size(); i++) HIP_CHECK(hipStreamWaitEvent(i ? streams[i - 1] : (hipStream_t) nullptr, se.second, 0));
                spare_events.push_back(se.second);
            }
            b.pending.clear();
            void *p = b.p;
            live[p] = it->first;
            free_blocks.erase(it);
            return p;
        }
#endif
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { // release the cache and retry once
            (void)hipGetLastError();
            release_locked();
            HIP_CHECK(hipMalloc(&p, bytes));
        }
        live[p] = bytes;
        return p;
    }
    void put(void *p) {
        std::lock_guard<std::mutex> g(mu);
        auto it = live.find(p);
        if (it == live.end()) { (void)hipFree(p); return; } // not ours (defensive)
        Block b{p, {}};
#ifndef TROYHIP_CPU_EMUL
        // With no stream registered the library knows ONE stream, the default one: whoever gets the block next uses it on that same stream, behind
        // everything that still touches it -- stream order is the whole guarantee, no free-point event is needed (an event record is a barrier packet
        // in the queue: 3-6 us between the kernels of consecutive operations, visible at one ciphertext per call).
        if (streams.empty()) { free_blocks.emplace(it->second, std::move(b)); live.erase(it); return; }
        // a registered stream that was destroyed without troyhip_stream_unregister must not break free() for good: its record fails, the stream is
        // dropped from the list (nothing can be running on it any more) and the block is cached all the same
        for (size_t i = 0; i <= streams.size();) {
            hipEvent_t e = new_event();
            hipStream_t st = i ? streams[i - 1] : (hipStream_t) nullptr;
            if (hipEventRecord(e, st) != hipSuccess) {
                (void)hipGetLastError();
                spare_events.push_back(e);
                if (i) { streams.erase(streams.begin() + (long)(i - 1)); g_stream_epoch.fetch_add(1, std::memory_order_release); continue; }
                // the default stream itself refuses: fall back to a device-wide wait, after which nothing is pending
                (void)hipDeviceSynchronize();
                for (auto &se : b.pending) spare_events.push_back(se.second);
                b.pending.clear();
                break;
            }
            b.pending.emplace_back(st, e);
            i++;
        }
#endif
        free_blocks.emplace(it->second, std::move(b));
        live.erase(it);
    }
    void release_locked() { // callers have synchronised the device (or are out of memory: hipFree synchronises)
        for (auto &kv : free_blocks) {
            for (auto &se : kv.second.pending) spare_events.push_back(se.second);
            (void)hipFree(kv.second.p);
        }
        free_blocks.clear();
    }
    void release() { std::lock_guard<std::mutex> g(mu); release_locked(); }
    void add_stream(hipStream_t s) {
        std::lock_guard<std::mutex> g(mu);
        if (!s || std::find(streams.begin(), streams.end(), s) != streams.end()) return;
#ifndef TROYHIP_CPU_EMUL
        if (streams.empty()) { // blocks cached so far carry no free-point event (put): from now on another stream may get them, so they wait for the default stream's work up to here
            for (auto &kv : free_blocks) {
                hipEvent_t e = new_event();
                if (hipEventRecord(e, nullptr) != hipSuccess) { (void)hipGetLastError(); spare_events.push_back(e); (void)hipDeviceSynchronize(); break; }
                kv.second.pending.emplace_back((hipStream_t) nullptr, e);
            }
        }
#endif
        streams.push_back(s);
    }
    void remove_stream(hipStream_t s) { // the caller has synchronised `s`: exactly its free-point events are complete, and they go with it
        std::lock_guard<std::mutex> g(mu);
        for (auto &kv : free_blocks) {
            auto &pd = kv.second.pending;
            for (size_t i = pd.size(); i-- > 0;)
                if (pd[i].first == s) { spare_events.push_back(pd[i].second); pd.erase(pd.begin() + (long)i); }
        }
        streams.erase(std::remove(streams.begin(), streams.end(), s), streams.end());
    }
    // one pool per device: instance() is the calling thread's current device's; a block is returned to the pool it came from (owner_of)
    static constexpr int kMaxDevices = 32;
    static DevicePool &of(int device) { static DevicePool pools[kMaxDevices]; return pools[(unsigned)device % kMaxDevices]; }
    static int current() { int d = 0; if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; } return d; }
    static DevicePool &instance() { return of(current()); }
    bool owns(void *p) { std::lock_guard<std::mutex> g(mu); return live.count(p) != 0; }
    static int owner_of(void *p) { // the current device first (the usual case), then the others
        const int cur = current();
        if (of(cur).owns(p)) return cur;
        for (int d = 0; d < kMaxDevices; d++) if (d != cur && of(d).owns(p)) return d;
        return cur;
    }
};
// troyhip_free from a thread that is bound to another device than the block's: events are recorded on the OWNER's streams, which needs the owner current
struct DeviceScope {
    int saved, target;
    explicit DeviceScope(int device) : saved(DevicePool::current()), target(device) { if (saved != target) HIP_CHECK(hipSetDevice(target)); }
    ~DeviceScope() { if (saved != target) (void)hipSetDevice(saved); }
};
} // namespace
int troyhip_malloc(void **out, size_t bytes) { return guard([&] { if (!out) throw Error(ST_INVALID_ARGUMENT, "null"); *out = DevicePool::instance().get(bytes ? bytes : 8); }); }
int troyhip_free(void *p) {
    return guard([&] {
        if (!p) return;
        const int owner = DevicePool::owner_of(p);
        DeviceScope scope(owner);
        DevicePool::of(owner).put(p);
    });
}
int troyhip_pool_release(void) { return guard([&] { HIP_CHECK(hipDeviceSynchronize()); DevicePool::instance().release(); }); }
// A caller's stream the library was never told about (a hipStream_t made elsewhere, a torch / RCCL stream) is announced to the pool the first time an
// entry point is handed it, so a block freed while that stream still reads it is not reused early (round-4 advisor: the pool records no free-point
// events until a stream is registered).  One compare per call; a stream that was unregistered is announced again when it comes back.
static inline hipStream_t on(void *stream) {
    if (stream) {
        // the streams this thread has announced in the current epoch: four entries, so that a caller alternating between a few streams (two lanes,
        // a copy stream) takes neither the pool's mutex nor its linear search per call
        static thread_local void *seen[4] = {nullptr, nullptr, nullptr, nullptr};
        static thread_local unsigned next = 0;
        static thread_local uint64_t last_epoch = ~0ull;
        const uint64_t epoch = g_stream_epoch.load(std::memory_order_acquire);
        if (epoch != last_epoch) { seen[0] = seen[1] = seen[2] = seen[3] = nullptr; last_epoch = epoch; }
        if (stream != seen[0] && stream != seen[1] && stream != seen[2] && stream != seen[3]) {
            DevicePool::instance().add_stream((hipStream_t)stream);
            seen[next] = stream;
            next = (next + 1) & 3u;
        }
    }
    return (hipStream_t)stream;
}
int troyhip_copy_h2d(void *dst, const void *src, size_t bytes, void *stream) {
    return guard([&] { HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream)); HIP_CHECK(hipStreamSynchronize((hipStream_t)stream)); });
}
int troyhip_copy_d2h(void *dst, const void *src, size_t bytes, void *stream) {
    return guard([&] { HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream)); HIP_CHECK(hipStreamSynchronize((hipStream_t)stream)); });
}
// direct access between two devices (xGMI), switched on the first time a pair is used: without it hipMemcpyPeerAsync still works, staged through host memory
static void enable_peer_access(int device, int peer) {
    static std::mutex mu;
    static std::vector<std::pair<int, int>> done;
    std::lock_guard<std::mutex> g(mu);
    if (std::find(done.begin(), done.end(), std::make_pair(device, peer)) != done.end()) return;
    done.emplace_back(device, peer);
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, device, peer) != hipSuccess || !can) { (void)hipGetLastError(); return; }
    DeviceScope scope(device);
    if (hipDeviceEnablePeerAccess(peer, 0) != hipSuccess) (void)hipGetLastError(); // "already enabled" (another library did it) is fine
}
int troyhip_copy_peer(void *dst, int dst_device, const void *src, int src_device, size_t bytes, void *stream) {
    return guard([&] {
        if (!bytes) return;
        if (dst_device == src_device) { DeviceScope scope(src_device); HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, on(stream))); return; }
        enable_peer_access(dst_device, src_device);
        enable_peer_access(src_device, dst_device);
        HIP_CHECK(hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, (hipStream_t)stream));
    });
}
int troyhip_copy_d2d(void *dst, const void *src, size_t bytes, void *stream) {
    return guard([&] { HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, on(stream))); });
}
int troyhip_memset_zero(void *dst, size_t bytes, void *stream) { return guard([&] { HIP_CHECK(hipMemsetAsync(dst, 0, bytes, on(stream))); }); }
int troyhip_stream_synchronize(void *stream) { return guard([&] { HIP_CHECK(hipStreamSynchronize((hipStream_t)stream)); }); }
int troyhip_stream_create(void **stream) {
    return guard([&] { hipStream_t st; HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); DevicePool::instance().add_stream(st); *stream = (void *)st; });
}
int troyhip_stream_register(void *stream) { return guard([&] { DevicePool::instance().add_stream((hipStream_t)stream); }); }
int troyhip_stream_unregister(void *stream) {
    return guard([&] {
        HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        DevicePool::instance().remove_stream((hipStream_t)stream);
        g_stream_epoch.fetch_add(1, std::memory_order_release);
    });
}
int troyhip_stream_destroy(void *stream) {
    return guard([&] {
        HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        DevicePool::instance().remove_stream((hipStream_t)stream);
        g_stream_epoch.fetch_add(1, std::memory_order_release);
        HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    });
}
int troyhip_device_pci_bus_id(int device, char *out, size_t capacity) {
    return guard([&] {
        if (!out || capacity < 16) throw Error(ST_INVALID_ARGUMENT, "pci bus id buffer");
#ifndef TROYHIP_CPU_EMUL
        HIP_CHECK(hipDeviceGetPCIBusId(out, (int)capacity, device));
#else
        (void)device;
        std::snprintf(out, capacity, "emul");
#endif
    }, false);
}
int troyhip_mem_info(size_t *free_bytes, size_t *total_bytes) { return guard([&] { HIP_CHECK(hipMemGetInfo(free_bytes, total_bytes)); }); }

/* device-side probes of the scalar modular arithmetic and the butterfly forms (selftest.hip); all pointers are DEVICE buffers */
int troyhip_test_modarith(int op, const uint64_t *a, const uint64_t *b, const uint64_t *c, uint64_t p, uint64_t aux, uint64_t *out, uint64_t n, void *stream) {
    return guard([&] {
        if (!a || !out || p < 2 || (p >> 61)) throw Error(ST_INVALID_ARGUMENT, "modarith probe arguments");
        launch_modarith_probe(op, a, b, c, p, aux, out, n, on(stream));
    });
}
/* per-kernel timing (rt.h): enable, run the launches to be measured, then fetch the report (JSON text, launch order; resets) */
int troyhip_stat(const char *name, uint64_t *value) {
    return guard([&] {
        if (!name || !value) throw Error(ST_INVALID_ARGUMENT, "null");
        for (int i = 0; i < stats::COUNT; i++)
            if (std::strcmp(name, stats::name(i)) == 0) { *value = stats::counter(i).load(std::memory_order_relaxed); return; }
        throw Error(ST_INVALID_ARGUMENT, std::string("no such counter: ") + name);
    }, false);
}
int troyhip_ktime_enable(int on) {
#ifndef TROYHIP_CPU_EMUL
    return guard([&] { ktime::enabled = on != 0; });
#else
    (void)on;
    return ST_OK;
#endif
}
int troyhip_ktime_report(char *out, size_t capacity) {
    return guard([&] {
        if (!out || !capacity) throw Error(ST_INVALID_ARGUMENT, "report buffer");
#ifndef TROYHIP_CPU_EMUL
        const std::string r = ktime::report();
#else
        const std::string r = "[]";
#endif
        if (r.size() + 1 > capacity) throw Error(ST_OUT_OF_RANGE, "report buffer too small");
        std::memcpy(out, r.c_str(), r.size() + 1);
    });
}
int troyhip_timer_create(void **timer) {
    return guard([&] {
        Timer *t = new Timer();
        HIP_CHECK(hipEventCreate(&t->a));
        HIP_CHECK(hipEventCreate(&t->b));
        *timer = t;
    });
}
int troyhip_timer_destroy(void *timer) { return guard([&] { Timer *t = (Timer *)timer; if (t) { (void)hipEventDestroy(t->a); (void)hipEventDestroy(t->b); delete t; } }); }
int troyhip_timer_start(void *timer, void *stream) { return guard([&] { HIP_CHECK(hipEventRecord(((Timer *)timer)->a, on(stream))); }); }
int troyhip_timer_stop(void *timer, void *stream) { return guard([&] { HIP_CHECK(hipEventRecord(((Timer *)timer)->b, on(stream))); }); }
int troyhip_timer_elapsed_ms(void *timer, float *ms) {
    return guard([&] { Timer *t = (Timer *)timer; HIP_CHECK(hipEventSynchronize(t->b)); HIP_CHECK(hipEventElapsedTime(ms, t->a, t->b)); });
}

int troyhip_coeff_modulus_create(uint64_t N, const int *bit_sizes, int count, uint64_t *out) {
    return guard([&] {
        if (!bit_sizes || !out || count < 1 || count > 64) throw Error(ST_INVALID_ARGUMENT, "bit_sizes is invalid");
        auto v = host::coeff_modulus_create(N, std::vector<int>(bit_sizes, bit_sizes + count));
        std::copy(v.begin(), v.end(), out);
    }, false);
}
int troyhip_plain_modulus_batching(uint64_t N, int bit_size, uint64_t *out) {
    return guard([&] { *out = host::coeff_modulus_create(N, {bit_size})[0]; }, false);
}

int troyhip_context_create(int scheme, uint64_t N, const uint64_t *coeff_modulus, int K, uint64_t plain_modulus, troyhip_context **out) {
    return guard([&] {
        if (!coeff_modulus || !out || K < 1) throw Error(ST_INVALID_ARGUMENT, "coeff_modulus is invalid");
        *out = new troyhip_context(scheme, N, std::vector<u64>(coeff_modulus, coeff_modulus + K), plain_modulus);
    });
}
int troyhip_context_create_host(int scheme, uint64_t N, const uint64_t *coeff_modulus, int K, uint64_t plain_modulus, troyhip_context **out) {
    return guard([&] {
        if (!coeff_modulus || !out || K < 1) throw Error(ST_INVALID_ARGUMENT, "coeff_modulus is invalid");
        *out = new troyhip_context(scheme, N, std::vector<u64>(coeff_modulus, coeff_modulus + K), plain_modulus, false);
    }, false);
}
int troyhip_context_destroy(troyhip_context *ctx) {
    return guard([&] {
        if (!ctx) return;
        if (ctx->ctx.has_device && g_init.load()) { DeviceScope scope(ctx->ctx.device); delete ctx; } // its tables and scratch go back to its own device's pool
        else delete ctx;
    }, false);
}
int troyhip_context_device(const troyhip_context *ctx, int *device) {
    return guard([&] {
        if (!ctx || !device) throw Error(ST_INVALID_ARGUMENT, "null");
        *device = ctx->ctx.has_device ? ctx->ctx.device : -1;
    }, false);
}
int troyhip_context_info(const troyhip_context *ctx, troyhip_context_info_t *out) {
    return guard([&] {
        if (!ctx || !out) throw Error(ST_INVALID_ARGUMENT, "null");
        const Context &c = need(ctx)->ctx;
        out->scheme = c.scheme; out->poly_modulus_degree = c.N; out->key_limbs = c.K;
        out->first_limbs = c.first_limbs; out->last_limbs = c.last_limbs; out->plain_modulus = c.t;
    }, false);
}
int troyhip_context_behz_bases(const troyhip_context *ctx, int limbs, uint64_t *bsk_out, int *bsk_size, uint64_t *gamma) {
    return guard([&] {
        const host::RnsLevel &r = need(ctx)->ctx.level(limbs).rns;
        std::copy(r.Bsk.begin(), r.Bsk.end(), bsk_out);
        *bsk_size = (int)r.Bsk.size();
        *gamma = r.gamma;
    }, false);
}
int troyhip_context_ntt_tables(const troyhip_context *ctx, uint64_t prime, uint64_t *rop, uint64_t *rquo, uint64_t *iop, uint64_t *iquo,
                               uint64_t *inv_degree2, uint64_t *root) {
    return guard([&] {
        const Context &c = need(ctx)->ctx;
        const host::NttTable &t = c.tables[c.prime_id(prime)];
        for (u64 i = 0; i < c.N; i++) { rop[i] = t.root[i].op; rquo[i] = t.root[i].quo; iop[i] = t.iroot[i].op; iquo[i] = t.iroot[i].quo; }
        inv_degree2[0] = t.inv_n.op; inv_degree2[1] = t.inv_n.quo;
        *root = t.psi;
    }, false);
}
int troyhip_context_release_stream(troyhip_context *ctx) { return guard([&] { need(ctx)->ctx.arena.release_owner(); }); }
int troyhip_context_reserve_scratch(troyhip_context *ctx, size_t words) { return guard([&] { need(ctx)->ctx.arena.reset(); need(ctx)->ctx.arena.reserve(words); }); }
int troyhip_context_scratch_words(const troyhip_context *ctx, int op, int limbs, uint64_t batch, size_t *words) {
    return guard([&] {
        if (op == 0) *words = need(ctx)->ev.scratch_multiply(2, 2, limbs, batch);
        else *words = need(ctx)->ev.scratch_switch_key(limbs, batch) + 2 * batch * limbs * need(ctx)->ctx.N + 128;
    });
}
int troyhip_galois_elt_from_step(const troyhip_context *ctx, int step, uint32_t *out) {
    return guard([&] { *out = host::galois_elt_from_step(need(ctx)->ctx.N, step); }, false);
}

// ---- CPU-side key generation / encryption / decryption (hostcrypto.cpp); all buffers are HOST memory ----
int troyhip_host_keygen(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, uint64_t *secret_key, uint64_t *public_key) {
    return guard([&] {
        hostcrypto::Rng rng(seed_lo, seed_hi);
        hostcrypto::keygen_secret(need(ctx)->ctx, rng, secret_key);
        if (public_key) hostcrypto::keygen_public(need(ctx)->ctx, rng, secret_key, public_key);
    }, false);
}
int troyhip_host_relin_key(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, const uint64_t *secret_key, uint64_t *out) {
    return guard([&] {
        hostcrypto::Rng rng(seed_lo, seed_hi, 1);
        std::vector<u64> src((size_t)need(ctx)->ctx.K * need(ctx)->ctx.N);
        hostcrypto::relin_source(need(ctx)->ctx, secret_key, src.data());
        hostcrypto::keygen_kswitch(need(ctx)->ctx, rng, secret_key, src.data(), out);
    }, false);
}
int troyhip_host_galois_key(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, const uint64_t *secret_key, uint32_t galois_elt, uint64_t *out) {
    return guard([&] {
        hostcrypto::Rng rng(seed_lo, seed_hi, ((u64)2 << 32) | galois_elt);
        std::vector<u64> src((size_t)need(ctx)->ctx.K * need(ctx)->ctx.N);
        hostcrypto::galois_source(need(ctx)->ctx, secret_key, galois_elt, src.data());
        hostcrypto::keygen_kswitch(need(ctx)->ctx, rng, secret_key, src.data(), out);
    }, false);
}
int troyhip_host_encrypt_zero(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, const uint64_t *key, int symmetric, int limbs, uint64_t *ct_out) {
    return guard([&] {
        hostcrypto::Rng rng(seed_lo, seed_hi, (u64)(symmetric ? 7 : 6) << 32);
        if (symmetric) hostcrypto::encrypt_zero_symmetric(need(ctx)->ctx, rng, key, limbs, ct_out);
        else hostcrypto::encrypt_zero(need(ctx)->ctx, rng, key, limbs, ct_out);
    }, false);
}
int troyhip_host_kswitch_key(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, const uint64_t *secret_key, const uint64_t *new_key, uint64_t *out) {
    return guard([&] {
        hostcrypto::Rng rng(seed_lo, seed_hi, (u64)5 << 32);
        hostcrypto::keygen_kswitch(need(ctx)->ctx, rng, secret_key, new_key, out);
    }, false);
}
int troyhip_host_encrypt(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, const uint64_t *public_key, const uint64_t *plain,
                         uint64_t n_coeffs, int limbs, uint64_t *ct_out) {
    return guard([&] {
        hostcrypto::Rng rng(seed_lo, seed_hi, (u64)3 << 32);
        hostcrypto::encrypt(need(ctx)->ctx, rng, public_key, plain, n_coeffs, limbs, ct_out);
    }, false);
}
int troyhip_host_encrypt_symmetric(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, const uint64_t *secret_key, const uint64_t *plain,
                                   uint64_t n_coeffs, int limbs, uint64_t *ct_out) {
    return guard([&] {
        hostcrypto::Rng rng(seed_lo, seed_hi, (u64)4 << 32);
        hostcrypto::encrypt_symmetric(need(ctx)->ctx, rng, secret_key, plain, n_coeffs, limbs, ct_out);
    }, false);
}
int troyhip_host_encrypt_symmetric_seeded(const troyhip_context *ctx, uint64_t seed_lo, uint64_t seed_hi, uint64_t a_seed, const uint64_t *secret_key,
                                          const uint64_t *plain, uint64_t n_coeffs, int limbs, uint64_t *ct_out) {
    return guard([&] {
        hostcrypto::Rng rng(seed_lo, seed_hi, (u64)(plain ? 4 : 7) << 32);
        hostcrypto::encrypt_symmetric_seeded(need(ctx)->ctx, rng, a_seed, secret_key, plain, n_coeffs, limbs, ct_out);
    }, false);
}
int troyhip_host_expand_seed(const troyhip_context *ctx, uint64_t a_seed, int limbs, uint64_t *c1_out) {
    return guard([&] { hostcrypto::expand_seed(need(ctx)->ctx, a_seed, limbs, c1_out); }, false);
}
int troyhip_host_decrypt(const troyhip_context *ctx, const uint64_t *secret_key, const uint64_t *ct, int size, int limbs, int is_ntt_form,
                         uint64_t correction_factor, uint64_t *plain_out) {
    return guard([&] { hostcrypto::decrypt(need(ctx)->ctx, secret_key, ct, size, limbs, is_ntt_form != 0, correction_factor, plain_out); }, false);
}

int troyhip_host_batch_encode(const troyhip_context *ctx, const uint64_t *values, uint64_t count, uint64_t *plain_out) {
    return guard([&] { hostcrypto::batch_encode(need(ctx)->ctx, values, count, plain_out); }, false);
}
int troyhip_host_batch_decode(const troyhip_context *ctx, const uint64_t *plain, uint64_t n_coeffs, uint64_t *values_out) {
    return guard([&] { hostcrypto::batch_decode(need(ctx)->ctx, plain, n_coeffs, values_out); }, false);
}

int troyhip_ntt(troyhip_context *ctx, uint64_t *data, uint64_t rows, const uint64_t *row_primes, int period, int inner, int inverse, void *stream) {
    return guard([&] {
        Context &c = need(ctx)->ctx;
        launch_ntt(data, c.d_desc, map_from_primes(c, row_primes, period, inner), rows, c.logn, inverse != 0, on(stream));
    });
}
int troyhip_fill_uniform(troyhip_context *ctx, uint64_t *data, uint64_t rows, const uint64_t *row_primes, int period, int inner, uint64_t seed,
                         uint64_t row0, void *stream) {
    return guard([&] {
        Context &c = need(ctx)->ctx;
        launch_fill_uniform(data, c.d_desc, map_from_primes(c, row_primes, period, inner), c.logn, seed, row0, rows, on(stream));
    });
}

int troyhip_negate(troyhip_context *ctx, troyhip_ct *a, uint64_t batch, void *stream) {
    return guard([&] { CtBatch x = view(a); need(ctx)->ev.negate(x, batch, on(stream)); store(x, a); });
}
int troyhip_add(troyhip_context *ctx, troyhip_ct *a, const troyhip_ct *b, uint64_t batch, void *stream) {
    return guard([&] { CtBatch x = view(a); need(ctx)->ev.add_sub(x, view(b), batch, false, on(stream)); store(x, a); });
}
int troyhip_sub(troyhip_context *ctx, troyhip_ct *a, const troyhip_ct *b, uint64_t batch, void *stream) {
    return guard([&] { CtBatch x = view(a); need(ctx)->ev.add_sub(x, view(b), batch, true, on(stream)); store(x, a); });
}
int troyhip_multiply(troyhip_context *ctx, const troyhip_ct *a, const troyhip_ct *b, troyhip_ct *out, uint64_t batch, void *stream) {
    return guard([&] { CtBatch o = view(out); need(ctx)->ev.multiply(view(a), view(b), o, batch, on(stream)); store(o, out); });
}
int troyhip_relinearize(troyhip_context *ctx, troyhip_ct *ct, const uint64_t *relin_key, uint64_t batch, void *stream) {
    return guard([&] {
        if (!relin_key) throw Error(ST_INVALID_ARGUMENT, "not enough relinearization keys");
        CtBatch x = view(ct);
        need(ctx)->ev.relinearize(x, KsKey{relin_key}, batch, on(stream));
        store(x, ct);
    });
}
int troyhip_relinearize_keys(troyhip_context *ctx, troyhip_ct *ct, const uint64_t *const *relin_keys, int n_keys, uint64_t batch, void *stream) {
    return guard([&] {
        if (n_keys < 0 || n_keys > 14 || (n_keys && !relin_keys)) throw Error(ST_INVALID_ARGUMENT, "not enough relinearization keys");
        KsKey keys[14];
        for (int i = 0; i < n_keys; i++) keys[i] = KsKey{relin_keys[i]};
        CtBatch x = view(ct);
        need(ctx)->ev.relinearize(x, keys, n_keys, batch, on(stream));
        store(x, ct);
    });
}
int troyhip_relinearize_to(troyhip_context *ctx, const troyhip_ct *in, troyhip_ct *out, const uint64_t *const *relin_keys, int n_keys, uint64_t batch, void *stream) {
    return guard([&] {
        if (n_keys < 0 || n_keys > 14 || (n_keys && !relin_keys)) throw Error(ST_INVALID_ARGUMENT, "not enough relinearization keys");
        if (!in || !out) throw Error(ST_INVALID_ARGUMENT, "null ciphertext");
        KsKey keys[14];
        for (int i = 0; i < n_keys; i++) keys[i] = KsKey{relin_keys[i]};
        CtBatch o = view(out);
        need(ctx)->ev.relinearize_to(view(in), o, keys, n_keys, batch, on(stream));
        store(o, out);
    });
}
int troyhip_switch_key(troyhip_context *ctx, troyhip_ct *ct, const uint64_t *target, uint64_t target_batch_stride, const uint64_t *kswitch_key,
                       uint64_t batch, void *stream) {
    return guard([&] { CtBatch x = view(ct); need(ctx)->ev.switch_key(x, target, target_batch_stride, KsKey{kswitch_key}, batch, on(stream)); store(x, ct); });
}
int troyhip_mod_switch_to_next(troyhip_context *ctx, const troyhip_ct *in, troyhip_ct *out, uint64_t batch, void *stream) {
    return guard([&] { CtBatch o = view(out); need(ctx)->ev.mod_switch_to_next(view(in), o, batch, on(stream)); store(o, out); });
}
int troyhip_rescale_to_next(troyhip_context *ctx, const troyhip_ct *in, troyhip_ct *out, uint64_t batch, void *stream) {
    return guard([&] { CtBatch o = view(out); need(ctx)->ev.rescale_to_next(view(in), o, batch, on(stream)); store(o, out); });
}
int troyhip_apply_galois(troyhip_context *ctx, troyhip_ct *ct, uint32_t galois_elt, const uint64_t *galois_key, uint64_t batch, void *stream) {
    return guard([&] { CtBatch x = view(ct); need(ctx)->ev.apply_galois(x, galois_elt, KsKey{galois_key}, batch, on(stream)); store(x, ct); });
}

static void rotate_internal(troyhip_context *ctx, CtBatch &x, int steps, const uint32_t *elts, const uint64_t *const *keys, int n_keys, u64 batch, hipStream_t s) {
    if (steps == 0) return; // evaluator_cuda.cu:2140-2143
    const u64 N = need(ctx)->ctx.N;
    auto find = [&](uint32_t e) -> const uint64_t * { for (int i = 0; i < n_keys; i++) if (elts[i] == e) return keys[i]; return nullptr; };
    uint32_t elt = host::galois_elt_from_step(N, steps);
    if (const uint64_t *k = find(elt)) { need(ctx)->ev.apply_galois(x, elt, KsKey{k}, batch, s); return; }
    auto ns = host::naf(steps); // evaluator_cuda.cu:2152-2175
    if (ns.size() == 1) throw Error(ST_INVALID_ARGUMENT, "Galois key not present");
    for (int st : ns) if ((u64)std::abs(st) != (N >> 1)) rotate_internal(ctx, x, st, elts, keys, n_keys, batch, s);
}
int troyhip_rotate(troyhip_context *ctx, troyhip_ct *ct, int steps, int conjugate, const uint32_t *key_elts, const uint64_t *const *keys, int n_keys,
                   uint64_t batch, void *stream) {
    return guard([&] {
        CtBatch x = view(ct);
        if (conjugate) { // rotateColumns / complexConjugate: element 2N-1 (evaluator_cuda.cuh:399-420)
            uint32_t elt = (uint32_t)(2 * need(ctx)->ctx.N - 1);
            const uint64_t *k = nullptr;
            for (int i = 0; i < n_keys; i++) if (key_elts[i] == elt) k = keys[i];
            if (!k) throw Error(ST_INVALID_ARGUMENT, "Galois key not present");
            need(ctx)->ev.apply_galois(x, elt, KsKey{k}, batch, on(stream));
        } else {
            rotate_internal(ctx, x, steps, key_elts, keys, n_keys, batch, on(stream));
        }
        store(x, ct);
    });
}
int troyhip_transform_to_ntt(troyhip_context *ctx, troyhip_ct *ct, uint64_t batch, void *stream) {
    return guard([&] { CtBatch x = view(ct); need(ctx)->ev.transform_to_ntt(x, batch, on(stream)); store(x, ct); });
}
int troyhip_transform_from_ntt(troyhip_context *ctx, troyhip_ct *ct, uint64_t batch, void *stream) {
    return guard([&] { CtBatch x = view(ct); need(ctx)->ev.transform_from_ntt(x, batch, on(stream)); store(x, ct); });
}
int troyhip_multiply_plain_ntt(troyhip_context *ctx, troyhip_ct *ct, const uint64_t *plain, double plain_scale, uint64_t batch, void *stream) {
    return guard([&] { CtBatch x = view(ct); need(ctx)->ev.multiply_plain_ntt(x, plain, plain_scale, batch, on(stream)); store(x, ct); });
}
int troyhip_multiply_plain_accumulate(troyhip_context *ctx, const troyhip_ct *const *cts, const uint64_t *const *plains, int count, double plain_scale, troyhip_ct *out,
                                      uint64_t batch, void *stream) {
    return guard([&] {
        if (!cts || !plains || count < 1 || count > 16) throw Error(ST_INVALID_ARGUMENT, "multiply_plain_accumulate takes 1 to 16 products");
        CtBatch v[16];
        for (int i = 0; i < count; i++) v[i] = view(cts[i]);
        CtBatch o = view(out);
        need(ctx)->ev.multiply_plain_accumulate(v, plains, count, plain_scale, o, batch, on(stream));
        store(o, out);
    });
}
int troyhip_add_plain(troyhip_context *ctx, troyhip_ct *ct, const uint64_t *plain, uint64_t plain_coeff_count, uint64_t plain_batch_stride, double plain_scale,
                      int subtract, uint64_t batch, void *stream) {
    return guard([&] {
        CtBatch x = view(ct);
        need(ctx)->ev.add_plain(x, plain, plain_coeff_count, plain_batch_stride, plain_scale, subtract != 0, batch, on(stream));
        store(x, ct);
    });
}
int troyhip_multiply_plain(troyhip_context *ctx, troyhip_ct *ct, const uint64_t *plain, uint64_t plain_coeff_count, uint64_t plain_batch_stride, uint64_t batch,
                           void *stream) {
    return guard([&] { CtBatch x = view(ct); need(ctx)->ev.multiply_plain(x, plain, plain_coeff_count, plain_batch_stride, batch, on(stream)); store(x, ct); });
}
int troyhip_plain_to_ntt(troyhip_context *ctx, const uint64_t *plain, uint64_t plain_coeff_count, uint64_t plain_batch_stride, int limbs, uint64_t *out,
                         uint64_t count, void *stream) {
    return guard([&] { need(ctx)->ev.plain_to_ntt(plain, plain_coeff_count, plain_batch_stride, limbs, out, count, on(stream)); });
}
int troyhip_apply_key_switching(troyhip_context *ctx, troyhip_ct *ct, const uint64_t *kswitch_key, uint64_t batch, void *stream) {
    return guard([&] { CtBatch x = view(ct); need(ctx)->ev.apply_key_switching(x, KsKey{kswitch_key}, batch, on(stream)); store(x, ct); });
}
int troyhip_negacyclic_shift(troyhip_context *ctx, troyhip_ct *ct, uint64_t shift, uint64_t batch, void *stream) {
    return guard([&] { CtBatch x = view(ct); need(ctx)->ev.negacyclic_shift(x, shift, batch, on(stream)); store(x, ct); });
}
int troyhip_divide_by_poly_modulus_degree(troyhip_context *ctx, troyhip_ct *ct, uint64_t mul, uint64_t batch, void *stream) {
    return guard([&] { CtBatch x = view(ct); need(ctx)->ev.divide_by_degree(x, mul, batch, on(stream)); store(x, ct); });
}
int troyhip_random_bytes(void *out, size_t n) { // the reference seeds its PRNG from std::random_device (src/randomgen.cpp:23,72)
    return guard([&] {
        unsigned char *p = static_cast<unsigned char *>(out);
        while (n) {
            const ssize_t got = getrandom(p, n, 0);
            if (got < 0) {
                if (errno == EINTR) continue;
                throw Error(ST_RUNTIME_ERROR, "getrandom failed");
            }
            p += got;
            n -= (size_t)got;
        }
    }, false);
}
int troyhip_blake2b(void *out, size_t outlen, const void *in, size_t inlen) {
    return guard([&] {
        if (!out || outlen == 0 || outlen > 64 || (!in && inlen)) throw Error(ST_INVALID_ARGUMENT, "blake2b");
        host::blake2b(out, outlen, in, inlen);
    }, false);
}
int troyhip_context_parms_id(const troyhip_context *ctx, int limbs, uint64_t out[4]) {
    return guard([&] {
        if (limbs < 1 || limbs > need(ctx)->ctx.K) throw Error(ST_INVALID_ARGUMENT, "parms_id is not valid for the current context");
        host::parms_id(need(ctx)->ctx.scheme, need(ctx)->ctx.N, need(ctx)->ctx.primes, limbs, need(ctx)->ctx.t, out);
    }, false);
}
int troyhip_decrypt(troyhip_context *ctx, const troyhip_ct *ct, const uint64_t *secret_key, uint64_t *plain_out, uint64_t plain_batch_stride, uint64_t batch,
                    void *stream) {
    return guard([&] { CtBatch x = view(ct); need(ctx)->ev.decrypt(x, secret_key, plain_out, plain_batch_stride, batch, on(stream)); });
}

} // extern "C"
