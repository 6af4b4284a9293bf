// device_types.h -- plain structs shared between host launch code and gfx950 kernels.
#pragma once
#include "modarith.h"

namespace troyhip {

// One registered prime with its NTT tables in HBM (replaces NTTTablesCuda, src/utils/ntt_cuda.cuh:81-100).
struct PrimeDesc {
    u64 p, cr0, cr1, two_p;
    Shoup inv_n;             // N^-1 mod p
    Shoup iroot_last_scaled; // psi^-? of the last inverse stage pre-multiplied by N^-1
    Shoup r64;               // 2^64 mod p (folds the high word of a 128-bit sum: reduce128 in behz.hip)
    const Shoup *root;       // [N] forward twiddles, bit-reversed order (src/utils/ntt.cpp:38-43)
    const Shoup *iroot;      // [N] inverse twiddles, scrambled order   (src/utils/ntt.cpp:49-54)
    Shoup aux;               // free for derived tables (Context::d_desc_md: q_special^-1 mod p); appended last -- behz.hip indexes the words above
    // primes below 2^50 only (else nullptr): the same tables as pairs of doubles (w, w / p) for the FP64 butterflies (fpmod.h), as bit patterns
    const Shoup *root_fp;
    const Shoup *iroot_fp;
    Shoup inv_n_fp, iroot_last_scaled_fp; // (w, w / p) of inv_n / iroot_last_scaled as they stand in THIS descriptor (the derived tables rescale them)
    // round 5: the per-lane rounds of the FP64 single-pass kernels use w alone (the quotient comes from 1 / p), so they read COMPACT tables of the doubles
    // w (8-byte entries, bit patterns): half the bytes and the cache footprint of the pair tables, and the 4 / 2 consecutive entries a thread needs are
    // one or two 16-byte loads.  root_w[j] = double(root[j].op);  iroot_w[j] = double(iroot[j + 1].op) -- shifted by the "+ 1" every inverse stage offset
    // carries (src/utils/ntt.cpp:49-54), so that the groups of 4 and 2 are 32- and 16-byte aligned.  nullptr with root_fp.
    const u64 *root_w;
    const u64 *iroot_w;
};
__host__ __device__ inline Mod mod_of(const PrimeDesc &d) { return Mod{d.p, d.cr0, d.cr1}; }

// prime id of limb-row r of a buffer = id[(r / inner) % period]
struct LimbMap {
    uint8_t id[64];
    uint32_t period, inner;
    uint64_t lean; // bit i: the prime of slot i lies in [2^33, 2^58) -- guard-free butterflies apply (bfly.h); host-side dispatch only
    uint64_t fp;   // bit i: the prime of slot i lies below 2^50 and has FP64 tables -- the FP64 instances apply (fpmod.h); host-side dispatch only
    const uint64_t *host_primes; // HOST pointer (never dereferenced on the device): the context's prime registry, indexed by id[] -- the launchers
                                 // walk the FP64 value bounds with it; nullptr: integer kernels only
};

#define TROY_BUF_OOB 0x80000000u // an offset no buffer range reaches (ranges are < 2^31 bytes)
#ifdef TROYHIP_CPU_EMUL
#define TROY_WAVE_SYNC() hip_emul::park(2) /* all live lanes of the wave arrive before any proceeds */
#define TROY_DYN_LDS(type, name) static type name[160 * 1024 / sizeof(type)]
// LDS-DMA: lane l of the wave copies 16 bytes from its own global address to (wave-uniform LDS base) + 16*l
#define TROY_GLDS16(gptr, lds_base) memcpy((char *)(lds_base) + 16 * (threadIdx.x & 63), (const void *)(gptr), 16)
#define TROY_GLDS16_POL(gptr, lds_base, pol) TROY_GLDS16(gptr, lds_base)
// 32x32x32 int8 matrix-core product: c[16] += A-fragment x B-fragment (16 signed bytes per lane each)
struct MfmaFrag { int8_t bytes[16]; };
struct MfmaAcc { int32_t v[16]; };
#define TROY_MFMA_I8(fa, fb, fc) hip_emul::mfma_i32_32x32x32_i8((fa).bytes, (fb).bytes, (fc).v)
inline void frag_set_word(MfmaFrag &f, int pos, u64 v) { memcpy(f.bytes + 8 * pos, &v, 8); } // bytes 8 pos .. 8 pos + 7 of the fragment
// bounds-checked buffer access: an offset beyond the range loads 0 / stores nothing, WITHOUT a branch
struct BufRsrc { char *base; u32 bytes; };
inline BufRsrc make_rsrc(const void *p, u32 bytes) { return BufRsrc{(char *)p, bytes}; }
inline u64 buf_load_u64(BufRsrc r, u32 off) { u64 v = 0; if ((u64)off + 8 <= r.bytes) memcpy(&v, r.base + off, 8); return v; }
inline void buf_store_u64(BufRsrc r, u32 off, u64 v) { if ((u64)off + 8 <= r.bytes) memcpy(r.base + off, &v, 8); }
#define TROY_WAIT_VMEM() hip_emul::park(2) /* lanes run one after another here: every lane's copy must have happened */
#define TROY_WAIT_LDS() hip_emul::park(2)  /* ... and every lane must have read before any lane overwrites */
#else
// global_load_lds_dwordx4 (gfx950): no VGPR round trip, completion is counted by vmcnt; hipcc does NOT order later LDS
// reads after it, so every consumer waits explicitly (TROY_WAIT_VMEM) and every overwrite of a staging buffer is
// issued only after the reads of its previous content returned (TROY_WAIT_LDS).
struct MfmaFrag { int bytes __attribute__((ext_vector_type(4))); };
struct MfmaAcc { int v __attribute__((ext_vector_type(16))); };
#define TROY_MFMA_I8(fa, fb, fc) ((fc).v = __builtin_amdgcn_mfma_i32_32x32x32_i8((fa).bytes, (fb).bytes, (fc).v, 0, 0, 0))
__device__ __forceinline__ void frag_set_word(MfmaFrag &f, int pos, u64 v) { // bytes 8 pos .. 8 pos + 7 of the fragment
    f.bytes[2 * pos] = (int)(u32)v;
    f.bytes[2 * pos + 1] = (int)(u32)(v >> 32);
}
// raw buffer access through a V# (gfx9 encoding 0x00020000: 32-bit data format, no swizzle): the hardware drops lanes whose
// offset is outside [0, bytes), so ragged edges cost no branch -- and straight-line code lets the compiler count the
// outstanding stores exactly instead of draining them (s_waitcnt vmcnt(0)) before every use of a prefetched load
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef unsigned int troy_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ BufRsrc make_rsrc(const void *p, u32 bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ u64 buf_load_u64(BufRsrc r, u32 off) {
    const troy_v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
    return (u64)v.x | ((u64)v.y << 32);
}
#ifndef TROY_BUF_ST_POL
#define TROY_BUF_ST_POL 0 // probe: cache policy of the buffer stores (BEHZ kernels): 2 = non-temporal
#endif
__device__ __forceinline__ void buf_store_u64(BufRsrc r, u32 off, u64 v) {
    troy_v2u d;
    d.x = (u32)v;
    d.y = (u32)(v >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(d, r, (int)off, 0, TROY_BUF_ST_POL);
}
#define TROY_GLDS16(gptr, lds_base)                                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr), (__attribute__((address_space(3))) void *)(lds_base), 16, 0, 0)
// the same with a cache-policy immediate (gfx940 encoding: 1 = sc0, 2 = nt, 16 = sc1); probes only (N1_DMA_POL, N2_DMA_POL)
#define TROY_GLDS16_POL(gptr, lds_base, pol)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr), (__attribute__((address_space(3))) void *)(lds_base), 16, 0, pol)
#define TROY_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define TROY_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// LDS operations of one wave execute in order, so intra-wave exchange needs no s_barrier; this only stops the
// compiler from moving LDS accesses across the exchange point
#define TROY_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#define TROY_DYN_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#endif

} // namespace troyhip
