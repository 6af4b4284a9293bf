// hip_emul.h -- TEST-ONLY single-threaded SIMT emulator used to compile troy_amd/csrc/*.hip for the
// host (g++ -DTROYHIP_CPU_EMUL).  Purpose: debug kernels and run sanitizers in a container without a
// GPU ("run sanitizers on the CPU build only").  It is NOT a product path: troy_amd never loads the
// emulated library, and GPU parity is established by the `-m gpu` tests on real hardware.
//
// Model: one block at a time; every HIP thread of the block is a ucontext fiber.  __syncthreads()
// parks a fiber until all live fibers of the block are parked at a barrier; wave-level exchanges
// (__shfl*) park until all live lanes of the 64-wide wave arrived.  Wave size is 64 as on gfx950.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <ucontext.h>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define __HIP_DEVICE_COMPILE__ 0

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) ulonglong2 { unsigned long long x, y; };
typedef int hipError_t;
typedef void *hipStream_t;
struct hipEmulEvent { std::chrono::steady_clock::time_point t; };
typedef hipEmulEvent *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

namespace hip_emul {

struct Fiber {
    ucontext_t ctx;
    char *stack = nullptr;
    int state = 0; // 0 runnable, 1 at block barrier, 2 at wave exchange, 3 done
};
struct State {
    dim3 threadIdx, blockIdx, blockDim, gridDim;
    ucontext_t sched;
    std::vector<Fiber> fibers;
    int cur = -1;
    const std::function<void()> *body = nullptr;
    uint64_t wave_buf[64 * 64]; // [wave][lane]
    int8_t mfma_buf[64 * 64][32]; // [wave][lane]: 16 A bytes, 16 B bytes
};
inline State &S() {
    static State s;
    return s;
}
static const size_t kStack = 256 * 1024;

inline void fiber_entry() {
    State &s = S();
    (*s.body)();
    s.fibers[s.cur].state = 3;
    swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}
inline void set_tid(int t) {
    State &s = S();
    s.threadIdx.x = t % s.blockDim.x;
    s.threadIdx.y = (t / s.blockDim.x) % s.blockDim.y;
    s.threadIdx.z = t / (s.blockDim.x * s.blockDim.y);
}
inline void park(int st) {
    State &s = S();
    int me = s.cur;
    s.fibers[me].state = st;
    swapcontext(&s.fibers[me].ctx, &s.sched);
    set_tid(me);
}
inline void run_block(int nthreads) {
    State &s = S();
    if ((int)s.fibers.size() < nthreads) {
        size_t old = s.fibers.size();
        s.fibers.resize(nthreads);
        for (size_t i = old; i < s.fibers.size(); i++) s.fibers[i].stack = (char *)malloc(kStack);
    }
    for (int t = 0; t < nthreads; t++) {
        Fiber &f = s.fibers[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
        f.state = 0;
    }
    int done = 0;
    while (done < nthreads) {
        bool progressed = false;
        for (int t = 0; t < nthreads; t++) {
            Fiber &f = s.fibers[t];
            if (f.state != 0) continue;
            s.cur = t;
            set_tid(t);
            swapcontext(&s.sched, &f.ctx);
            progressed = true;
            if (f.state == 3) done++;
        }
        // release block barrier when every live fiber is parked at it
        int at_bar = 0, live = 0;
        for (int t = 0; t < nthreads; t++) { if (s.fibers[t].state != 3) live++; if (s.fibers[t].state == 1) at_bar++; }
        if (live && at_bar == live) { for (int t = 0; t < nthreads; t++) if (s.fibers[t].state == 1) s.fibers[t].state = 0; progressed = true; }
        // release wave exchanges wave by wave
        for (int w = 0; w * 64 < nthreads; w++) {
            int wl = 0, wa = 0;
            for (int t = w * 64; t < nthreads && t < (w + 1) * 64; t++) { if (s.fibers[t].state != 3) wl++; if (s.fibers[t].state == 2) wa++; }
            if (wl && wa == wl) { for (int t = w * 64; t < nthreads && t < (w + 1) * 64; t++) if (s.fibers[t].state == 2) s.fibers[t].state = 0; progressed = true; }
        }
        if (!progressed) { fprintf(stderr, "hip_emul: deadlock (divergent barrier)\n"); abort(); }
    }
}
inline std::mutex &launch_mutex() { static std::mutex m; return m; }
inline void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
    std::lock_guard<std::mutex> one_at_a_time(launch_mutex()); // the emulator has ONE set of fibers and static "LDS": host threads take turns, a launch runs to completion
    State &s = S();
    s.gridDim = grid;
    s.blockDim = block;
    s.body = &body;
    int nthreads = block.x * block.y * block.z;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                s.blockIdx = dim3(bx, by, bz);
                run_block(nthreads);
            }
}
inline int lane_id() { State &s = S(); return s.cur & 63; }
inline int wave_id() { State &s = S(); return s.cur >> 6; }
template <class T> inline T wave_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "wave_exchange");
    State &s = S();
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    s.wave_buf[wave_id() * 64 + lane_id()] = bits;
    park(2);
    uint64_t r = s.wave_buf[wave_id() * 64 + (src_lane & 63)];
    park(2); // everyone has read before anyone overwrites
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
// v_mfma_i32_32x32x32_i8 (layout verified on hardware by tools/mfma_probe.hip):
//   A: lane l = row l % 32, k = 16 (l / 32) + 0..15;  B: lane l = column l % 32, same k;  D reg r: row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32
inline void mfma_i32_32x32x32_i8(const int8_t (&a)[16], const int8_t (&b)[16], int32_t (&c)[16]) {
    State &s = S();
    const int w = wave_id(), l = lane_id();
    memcpy(s.mfma_buf[w * 64 + l], a, 16);
    memcpy(s.mfma_buf[w * 64 + l] + 16, b, 16);
    park(2);
    for (int r = 0; r < 16; r++) {
        const int row = 8 * (r / 4) + 4 * (l / 32) + r % 4, col = l % 32;
        int32_t acc = 0;
        for (int k = 0; k < 32; k++)
            acc += (int32_t)s.mfma_buf[w * 64 + row + 32 * (k / 16)][k % 16] * (int32_t)s.mfma_buf[w * 64 + col + 32 * (k / 16)][16 + k % 16];
        c[r] += acc;
    }
    park(2);
}
} // namespace hip_emul

#define threadIdx (hip_emul::S().threadIdx)
#define blockIdx (hip_emul::S().blockIdx)
#define blockDim (hip_emul::S().blockDim)
#define gridDim (hip_emul::S().gridDim)

inline void __syncthreads() { hip_emul::park(1); }
template <class T> inline T __shfl(T v, int src, int width = 64) { int l = hip_emul::lane_id(); return hip_emul::wave_exchange(v, (l & ~(width - 1)) | (src & (width - 1))); }
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) { int l = hip_emul::lane_id(); return hip_emul::wave_exchange(v, l ^ mask); }
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) { int l = hip_emul::lane_id(); int s = l + d; if ((s & ~(width - 1)) != (l & ~(width - 1))) s = l; return hip_emul::wave_exchange(v, s); }
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) { int l = hip_emul::lane_id(); int s = l - (int)d; if (s < (l & ~(width - 1))) s = l; return hip_emul::wave_exchange(v, s); }

inline uint64_t __umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
inline uint32_t __brev(uint32_t x) { uint32_t r = 0; for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i); return r; }
template <class T> inline T __builtin_amdgcn_readfirstlane(T v) { return v; } // only used on wave-uniform values
template <class T> inline T atomicAdd(T *p, T v) { T o = *p; *p += v; return o; }

#define TROY_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    do { auto _k = [=]() { kernel(__VA_ARGS__); }; hip_emul::launch(dim3(grid), dim3(block), _k); } while (0)

// ---- host runtime API subset ----
inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
// virtual devices (HIP_EMUL_DEVICES, default 1): the current device is per host thread, as in HIP; all of them share the host's memory, so what
// the multi-device code paths of the library do with device ids (per-device pools, context binding, peer copies) runs for real
inline int hip_emul_device_count() { static const int n = [] { const char *e = getenv("HIP_EMUL_DEVICES"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v > 16 ? 16 : v; }(); return n; }
inline int &hip_emul_current_device() { static thread_local int d = 0; return d; }
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= hip_emul_device_count()) return hipErrorInvalidValue; hip_emul_current_device() = d; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = hip_emul_device_count(); return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = hip_emul_current_device(); return hipSuccess; }
inline hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { memmove(d, s, n); return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; } // a small "chip": the launch planner sees several rounds
inline hipError_t hipMalloc(void **p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipEmulEvent(); return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new hipEmulEvent(); return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; } // launches run to completion where they are issued
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = *t = size_t(8) << 30; return hipSuccess; }
