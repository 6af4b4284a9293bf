// rt.h -- runtime layer of libtroyhip: HIP on gfx950.
//
// The product build (hipcc --offload-arch=gfx950) uses the HIP runtime directly.  A second,
// TEST-ONLY build (-DTROYHIP_CPU_EMUL, see tests/emul/) compiles the very same kernel sources for
// the host with a fiber-based SIMT emulator so that kernels can be run under sanitizers and debugged
// in a container without a GPU.  The emulated library is never loaded by the troy_amd package
// (troy_amd/capi.py refuses it) -- it is not a fallback path.
#pragma once

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>

#ifdef TROYHIP_CPU_EMUL
#include "hip_emul.h"
#else
#include <hip/hip_runtime.h>
// Per-kernel timing on demand (troyhip_ktime_enable / troyhip_ktime_report, capi.cpp): when enabled, every launch is bracketed by
// HIP events on ITS OWN stream and accumulated under the kernel's name -- bench.py's roofline.per_kernel comes from here, live.
// Disabled (the default) it costs one predictable branch per launch.
#include <atomic>
namespace troyhip { namespace ktime {
extern std::atomic<bool> enabled;
extern thread_local const char *tag; // optional name for the next launch (template instances share one source text)
void begin(const char *name, hipStream_t s); // remembers the record it opened per THREAD: end() closes that one, whatever other threads launched meanwhile
void end(hipStream_t s);
} }
// An empty grid (a batch of zero ciphertexts, a transform of zero rows) is no work, not an error: HIP rejects it as an invalid configuration,
// so it is not launched at all.
#define TROY_LAUNCH(kernel, grid, block, shmem, stream, ...)                          \
    do {                                                                              \
        const dim3 troy_grid_ = dim3(grid);                                           \
        if (!troy_grid_.x || !troy_grid_.y || !troy_grid_.z) break;                   \
        if (::troyhip::ktime::enabled) ::troyhip::ktime::begin(#kernel, stream);      \
        kernel<<<troy_grid_, block, shmem, stream>>>(__VA_ARGS__);                    \
        if (::troyhip::ktime::enabled) ::troyhip::ktime::end(stream);                 \
    } while (0)
#endif

namespace troyhip {

// Environment switches.  The shipped library reads FIVE, each once (documented in DESIGN.md section 8): TROYHIP_NTT = single | twopass (which transform form
// takes a launch), TROYHIP_FP64 = off (integer kernels for every prime), TROYHIP_AUX_BASE = reference (the reference's 61-bit BEHZ base),
// TROYHIP_SMALL = split | merged (the merged forms of small launches never / always), TROYHIP_SYNC = 1 (every entry point returns with the device idle: capi.cpp).  Everything else -- forcing an unfused fallback, guarded
// butterflies, rows per workgroup -- exists only in probe builds (-DTROYHIP_PROBES: `make probes`, and the CPU emulator build of the test suite).
inline const char *probe_env(const char *name) {
#ifdef TROYHIP_PROBES
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// Path counters (troyhip_stat, include/troyhip.h): which kernel class a launcher chose.  The parity tests read them so that a test of
// the FP64 instances cannot pass on the integer kernels (or the other way round) without saying so.
namespace stats {
enum { KS_FP_LAUNCHES, KS_INT_LAUNCHES, NTT1_FP_LAUNCHES, NTT1_INT_LAUNCHES, NTT2_FP_LAUNCHES, NTT2_INT_LAUNCHES, BEHZ_FP_LAUNCHES, BEHZ_MFMA_LAUNCHES, BEHZ_VALU_LAUNCHES, NTT2_WIDE_LAUNCHES, COUNT };
inline std::atomic<uint64_t> &counter(int i) { static std::atomic<uint64_t> c[COUNT]; return c[i]; } // host threads launch concurrently (one context per thread)
inline const char *name(int i) { static const char *n[COUNT] = {"ks_fp_launches", "ks_int_launches", "ntt1_fp_launches", "ntt1_int_launches", "ntt2_fp_launches", "ntt2_int_launches", "behz_fp_launches", "behz_mfma_launches", "behz_valu_launches", "ntt2_wide_launches"}; return n[i]; }
}

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// status codes of the C ABI (include/troyhip.h)
enum { ST_OK = 0, ST_INVALID_ARGUMENT = 1, ST_LOGIC_ERROR = 2, ST_OUT_OF_RANGE = 3, ST_RUNTIME_ERROR = 4, ST_NOT_INITIALIZED = 5 };

inline void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw Error(ST_RUNTIME_ERROR, std::string("HIP error: ") + hipGetErrorString(e) + " in " + what);
}
#define HIP_CHECK(x) ::troyhip::hip_check((x), #x)

// checked after every launch (the reference never checks launches: SURVEY.md section 5)
inline void launch_check(const char *name) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw Error(ST_RUNTIME_ERROR, std::string("kernel launch failed: ") + name + ": " + hipGetErrorString(e));
}

inline unsigned ceil_div(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

} // namespace troyhip
