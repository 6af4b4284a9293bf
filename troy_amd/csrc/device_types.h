// device_types.h -- plain structs shared between host launch code and gfx950 kernels.
#pragma once
#include "modarith.h"

namespace troyhip {

// One registered prime with its NTT tables in HBM (replaces NTTTablesCuda, src/utils/ntt_cuda.cuh:81-100).
struct PrimeDesc {
    u64 p, cr0, cr1, two_p;
    Shoup inv_n;             // N^-1 mod p
    Shoup iroot_last_scaled; // psi^-? of the last inverse stage pre-multiplied by N^-1
    const Shoup *root;       // [N] forward twiddles, bit-reversed order (src/utils/ntt.cpp:38-43)
    const Shoup *iroot;      // [N] inverse twiddles, scrambled order   (src/utils/ntt.cpp:49-54)
};
__host__ __device__ inline Mod mod_of(const PrimeDesc &d) { return Mod{d.p, d.cr0, d.cr1}; }

// prime id of limb-row r of a buffer = id[(r / inner) % period]
struct LimbMap {
    uint8_t id[64];
    uint32_t period, inner;
};

#ifdef TROYHIP_CPU_EMUL
#define TROY_WAVE_SYNC() hip_emul::park(2) /* all live lanes of the wave arrive before any proceeds */
#define TROY_DYN_LDS(type, name) static type name[160 * 1024 / sizeof(type)]
#else
// LDS operations of one wave execute in order, so intra-wave exchange needs no s_barrier; this only stops the
// compiler from moving LDS accesses across the exchange point
#define TROY_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#define TROY_DYN_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#endif

} // namespace troyhip
