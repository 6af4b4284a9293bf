// ntt1.hip -- single-pass NTT / INTT for N = 2^15 on gfx950: ONE HBM round trip per limb-transform.
//
// Replaces the two launches of ntt2.hip (strided pass + contiguous pass, 32 B of HBM traffic per coefficient) for the plain
// transforms of the reference (kNttNegacyclicHarvey / kInverseNttNegacyclicHarvey, src/kernelutils.cuh:578-632, host loops
// src/kernelutils.cu:330-371, stage kernels :373-476, CPU twin src/utils/dwthandler.h:88-372).  Same transform, same tables,
// canonical outputs: any exact evaluation order is bit-identical.
//
// A limb is 256 KiB; a CU has 512 KiB of vector registers and 160 KiB of LDS.  One 1024-thread workgroup (16 waves, 4 per
// SIMD, <= 128 VGPRs) owns a whole limb:
//   * round A (forward: the first 5 stages, gaps 2^14 .. 2^10) runs on 32 coefficients per thread, straight from HBM
//     (512-byte contiguous runs per wave and instruction); every twiddle of these stages is workgroup-uniform (scalar loads);
//   * the limb then splits into 32 independent 1024-point sub-transforms.  They go through LDS in two halves of 16 sub-blocks
//     (128 KiB): each WAVE owns one sub-block of 1024 points = 16 per lane and runs its remaining 10 stages as radix-16,
//     radix-16, radix-4 rounds IN PLACE in its own 8 KiB region (no workgroup barrier inside; LDS operations of a wave
//     execute in order), with the XOR swizzle sw1() that is bank-conflict-free for every access pattern used (checked
//     exhaustively against the LDS lane-group rules of MI355X_MICROARCH.md: tools/lds_banks.py);
//   * the other half waits in registers (32 VGPRs) meanwhile; 4 workgroup barriers per limb in total;
//   * the next limb's loads are issued as soon as registers are free and the butterflies that do not need them run first
//     (forward: the radix-16 part of round A on the even registers; inverse: LDS-DMA staging during round A), so the CU
//     streams from HBM while it computes;
//   * the inverse runs the same schedule backwards (sub-blocks first, the cross-wave round last, N^-1 folded in).
// Butterflies: bfly.h.  LEAN = primes below 2^58 (everything CoeffModulus::Create makes for sizes <= 58 bits): the forward
// transform needs no range guard at all in 15 stages (bound 1 + 15 * 3 = 46 p < 2^64), 16 VALU instructions per butterfly
// instead of 20; other primes keep the guarded forms.
#include "kernels.h"
#include <atomic>
#include "bfly.h"
#include "fpmod.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace troyhip {

#define N1_THREADS 1024
#define N1_LOGN 15
#define N1_N (1u << N1_LOGN)

struct Ntt1Args {
    u64 *data;            // rows of N coefficients, transformed in place ...
    const u64 *src;       // ... or read from here when not null: same row layout, or (inverse, src_ostride != 0) outer index o at
    u64 src_ostride;      //     src + o * src_ostride with its limbs packed behind each other (a strided batch of ciphertext polynomials)
    const PrimeDesc *primes;
    LimbMap map;          // row r = (o * period + i) * inner + k  has prime map.id[i]
    unsigned m_total;     // outer * inner rows per prime slot
    unsigned rows_per_wg;
    unsigned chunks;      // ceil(m_total / rows_per_wg)
    unsigned nslots;      // a launch covers the prime slots slots[0 .. nslots) of the pattern (one launch per prime class)
    unsigned xcd_per;     // != 0: the grid is 8 * xcd_per workgroups and workgroup b takes unit (b % 8) * xcd_per + b / 8 of the slot-major list (n1_unit)
    unsigned xcd_group;   // > 1: that list is ordered (group of xcd_group primes, chunk, prime within the group)
    unsigned xcd_perturb; // probe builds only (TROYHIP_NTT1_XCD_PERTURB): a deliberately WRONG grouped order, for the test that must notice it
    uint8_t slots[64];
    // forward only: divide-and-round correction (Ntt1Corr): rows are BUILT from cr_last on load and COMBINED with cr_in on store
    const u64 *cr_last, *cr_in;
    u64 *cr_out;
    const Shoup *cr_inv;
    u64 cr_in_ostride, cr_in_gstride, cr_out_gstride, cr_out_ostride, cr_qx, cr_half;
    unsigned cr_group, cr_accumulate;
    const u64 *cr_base;   // accumulate onto (base, 0) per group instead of onto out (Ntt1Corr::base)
    u64 cr_base_gstride;
    int cr_base_polys, md_base_polys; // 2: the second member / component reads its base polynomial too (relinearize out of place)
    // inverse only: mod-down epilogue (Ntt1ModDown); md_ct == nullptr: plain stores
    u64 *md_ct;
    u64 md_ct_bstride, md_qk, md_half;
    const u64 *md_base;   // accumulate onto (base[b], 0) instead of onto ct (Ntt1ModDown::base)
    u64 md_base_bstride;
    unsigned md_dl;
    unsigned fp_red_mask; // FP64 instances (ntt1_*_fp_kernel, fpmod.h): bit r = reduce the values before round r (forward: A, B, C1, C2; inverse:
                          // D', C', B', A', the last stage of A'); walked on the host (fp_plan / fp_plan_inv)
    u64 *dbg;             // development builds (-DN1_TIMING): s_memtime stamps of wave 0 of workgroup dbg_block
    unsigned dbg_block;
};
// wave priority by progress: a wave that is ahead of the others in a barrier-to-barrier segment lowers its own priority, so the
// four waves of a SIMD advance together (the arbiter otherwise serves the oldest wave first and the segment ends with one wave
// per SIMD running alone)
#if !defined(N1_PRIO_OFF) && !defined(TROYHIP_CPU_EMUL) // forward kernels only; on since they store lane-linearly (round 4: integer forward -3 %, FP64 -1 %; before that neutral).  The same hints in the inverse kernels: FP64 +4 % SLOWER, integer neutral
#define N1_PRIO(k) __builtin_amdgcn_s_setprio(k)
#else
#define N1_PRIO(k)
#endif
#ifdef N1_TIMING
#define N1_STAMP(i) do { if (a.dbg && threadIdx.x == 0 && blockIdx.x == a.dbg_block) a.dbg[(i) + 16 * (mm - m_begin)] = clock64(); } while (0)
#else
#define N1_STAMP(i)
#endif

// position of element j (0..1023) of a sub-block inside its 8 KiB LDS region.  Involution; keeps bit 0 (16-byte pairs stay
// together).  Conflict-free for: j = 64 r + lane (rounds A/B, 8-byte accesses), j = 64 (lane / 4) + 4 r + lane % 4 (round C),
// j = 256 g + 4 lane + {0, 2} (round D, 16-byte reads), j = 128 i + 2 lane (lane-linear 16-byte staging).
__device__ __forceinline__ unsigned sw1(unsigned j) { return j ^ (((j >> 6) & 7u) << 2) ^ (((j >> 5) & 1u) << 1); }
// The inverse transform also WRITES 16 bytes per lane at j = 8 u + 2 q (its first round leaves a thread's eight consecutive coefficients);
// ds_write_b128 is served in groups of 8 consecutive lanes, and under sw1 lanes l and l + 2 of a group meet in one bank (2-way:
// SQ_LDS_BANK_CONFLICT = 34 % of the inverse kernel's LDS cycles in round 2, 0 in the forward kernel, which only reads that pattern).  One more
// term -- bit 2 ^= bit 4 -- separates them and keeps every other pattern conflict-free (tools/lds_banks.py); it is no involution any more
// (bit 4 is both a source and a target), so the LDS-DMA staging, which needs "which element belongs at position x", uses sw2_inv.
#ifndef N1_FP_FENCE_B
#define N1_FP_FENCE_B 2 // FP64 forward kernel, round B (16 values in registers next to the waiting half): butterflies in flight between scheduling fences
#endif
#ifndef N1_FWD_STORE_VIA_LDS
#define N1_FWD_STORE_VIA_LDS 1 // 0 (probe): the forward kernels store each thread's eight consecutive results directly
#endif
#ifndef N1_CR_LINEAR
#define N1_CR_LINEAR 1 // 0 (probe): the divide-and-round epilogue on a thread's eight consecutive coefficients (16-byte loads every 64 bytes)
#endif
#ifndef N1_FWD_STORE_UNROLL
#define N1_FWD_STORE_UNROLL 2
#endif
#ifndef N1_FP_ODD_EARLY
#define N1_FP_ODD_EARLY 0 // 1 (probe): the odd half of a row is converted up front, as before round 4
#endif
#ifndef N1_INV_SWZ
#define N1_INV_SWZ 2
#endif
#ifndef N1_INV_EXP
#define N1_INV_EXP 0 // removal probes of the N = 2^15 inverse body (wrong results; tools/ntt_probe.sh name:-DN1_INV_EXP=k): 1 no global reads, 2 no global writes, 4 no workgroup barriers
#endif
#ifndef N1_DMA_POL
#define N1_DMA_POL 0 // probe: cache policy of the inverse kernels' LDS-DMA staged half (2 = non-temporal)
#endif
#ifndef N1_NT_INV
#define N1_NT_INV 0 // non-temporal stores in the INTEGER N = 2^15 inverse kernels: bit 0 the plain rows, bit 1 the mod-down epilogue's
#endif
// workgroup -> unit of the slot-major list (slot, chunk).  The hardware deals consecutive workgroups to the 8 XCDs in turn; with the flat mapping every XCD
// therefore works on every prime in flight (2-4 at a time: each a 1 MiB twiddle table competing for that XCD's 4 MiB of L2 with the rows streaming through).
// XCD-aware: XCD x walks its own contiguous eighth of the list, one prime at a time.
#ifndef N1_XCD
#define N1_XCD 1
#endif
// xcd_group > 1 (the forms whose rows of DIFFERENT primes read one shared row: the special limb of a mod-down, the dropped limb of a divide-and-round): the
// list is ordered (group of xcd_group primes, chunk, prime within the group), so the workgroups an XCD starts back to back are the same rows under
// xcd_group primes and the shared row is fetched into that L2 once for all of them -- at the price of xcd_group twiddle tables in it.
#ifndef N1_XCD_GROUP_FP
#define N1_XCD_GROUP_FP 4
#endif
#ifndef N1_XCD_GROUP_INT
#define N1_XCD_GROUP_INT 4
#endif
__device__ __forceinline__ bool n1_unit(const Ntt1Args &a, unsigned &sidx, unsigned &chunk) {
    unsigned q = blockIdx.x;
    if (a.xcd_per) {
        q = (blockIdx.x & 7u) * a.xcd_per + (blockIdx.x >> 3);
        if (q >= a.nslots * a.chunks) return false; // the padding of the last eighth
    }
    if (!a.xcd_per || a.xcd_group <= 1) {
        sidx = q / a.chunks;
        chunk = q - sidx * a.chunks;
#ifdef TROYHIP_PROBES
        if (a.xcd_perturb && a.xcd_per && sidx == 1 && a.chunks > 3 && chunk == a.chunks / 2 + 1) chunk -= 1; // (see below; the XCD-aware flat list)
#endif
    } else {
        const unsigned gs = a.xcd_group * a.chunks, g = q / gs, r = q - g * gs, first = g * a.xcd_group;
        const unsigned gsz = a.nslots - first < a.xcd_group ? a.nslots - first : a.xcd_group; // the last group may be short
        chunk = r / gsz;
        sidx = first + (r - chunk * gsz);
#ifdef TROYHIP_PROBES
        // tests/test_gpu_parity.py::test_full_headline_batch_catches_a_perturbed_xcd_order: one interior chunk of the second group is mapped onto its
        // neighbour (in range: nothing faults; the chunk's rows are never transformed, its neighbour's are written twice with the same values)
        if (a.xcd_perturb && g == 1 && a.chunks > 3 && chunk == a.chunks / 2 + 1) chunk -= 1;
#endif
    }
    return true;
}
#ifndef N1_STAGGER
#define N1_STAGGER 0 // probe: the workgroups of a launch's first round start N1_STAGGER x 3.9 us x ((blockIdx / 8) % 8) late -- are the CUs of an XCD phase-aligned (bursty traffic)?
#endif
__device__ __forceinline__ void n1_stagger() {
#if N1_STAGGER && !defined(TROYHIP_CPU_EMUL)
    if (blockIdx.x < 256) {
        const unsigned ph = (blockIdx.x >> 3) & 7;
        for (unsigned i = 0; i < ph * N1_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
    }
#endif
}
#ifndef N1_INV_TOP
#define N1_INV_TOP 0 // where an inverse row waits for its LDS-DMA staged half: 0 at the top of the row (after requesting the second half); probes, all measured
#endif               // NEUTRAL in round 5 (profiles/r05_inv_probes.txt): 1 drain then request, 2 request then vmcnt(8), 3 wait BEFORE the previous row's stores

__device__ __forceinline__ unsigned sw2(unsigned j) { return N1_INV_SWZ == 2 ? sw1(j) ^ (((j >> 4) & 1u) << 2) : sw1(j); }
#ifndef N1_FWD_WB_SW1
#define N1_FWD_WB_SW1 0 // 1 (probe): the forward kernels' write-back for the lane-linear stores under sw1 (2-way conflicted 16-byte writes)
#endif
__device__ __forceinline__ unsigned wb_swz(unsigned j) { return N1_FWD_WB_SW1 ? sw1(j) : sw2(j); }
__device__ __forceinline__ unsigned sw2_inv(unsigned x) { // bits 5.. are untouched by sw2: undo bit 4 first, then the term it feeds
    if (N1_INV_SWZ != 2) return sw1(x);
    const unsigned y = sw1(x);
    return y ^ (((y >> 4) & 1u) << 2);
}

__device__ __forceinline__ u64 *lds_at(u64 *lds_base, unsigned byte_off) { return reinterpret_cast<u64 *>(reinterpret_cast<char *>(lds_base) + byte_off); }

// twiddle loads.  The table pointers come out of a PrimeDesc that was itself loaded from memory, so the compiler would use flat
// loads; the tables are read-only global memory: per-lane entries go through global loads, workgroup-/wave-uniform entries
// through the scalar cache (constant address space), 16 bytes each
#ifdef TROYHIP_CPU_EMUL
__device__ __forceinline__ Shoup ld_tw(const Shoup *base, unsigned idx) { return base[idx]; }
__device__ __forceinline__ Shoup ld_tw_uniform(const Shoup *p) { return *p; }
#else
typedef unsigned long long troy_v2ull __attribute__((ext_vector_type(2)));
#ifndef N1_EXP_NOTW
#define N1_EXP_NOTW 0 // removal probe (wrong results): 1 = the per-lane twiddles are made up from the index instead of loaded -- what their latency costs
#endif
__device__ __forceinline__ Shoup ld_tw(const Shoup *base, unsigned idx) { // base is wave-uniform (SGPR pair), idx per lane: saddr + 32-bit offset
    if (N1_EXP_NOTW) return Shoup{(u64)idx * 0x9E3779B97F4A7C15ull >> 8, (u64)idx * 0xD1B54A32D192ED03ull};
    const troy_v2ull v = ((const __attribute__((address_space(1))) troy_v2ull *)base)[idx];
    return Shoup{v.x, v.y};
}
__device__ __forceinline__ Shoup ld_tw_uniform(const Shoup *p) {
    const troy_v2ull v = *((const __attribute__((address_space(4))) troy_v2ull *)p);
    return Shoup{v.x, v.y};
}
#endif
// FP64 instances, per-lane twiddles: only the double w is needed -- half the twiddle registers of the sub-block rounds; the quotient then comes from the
// rounded product and 1 / p (fp_mulmod_pinv).  They come from the compact tables (PrimeDesc::root_w / iroot_w: 8-byte entries): one entry, or 2 / 4 consecutive ones as one / two 16-byte loads (idx a multiple of
// 2 / 4: the groups are aligned).  Round 5: a removal probe (N1_EXP_NOTW) put the per-lane twiddle loads at 10 % of the FP64 single-pass kernels (1 % of the
// integer ones, whose 15-instruction butterflies hide them): from the pair tables a thread's 4 + 2 + 1 twiddles were seven 8-byte loads at a 16-byte stride.
__device__ __forceinline__ Shoup ld_w1(const u64 *base, unsigned idx) {
#ifdef TROYHIP_CPU_EMUL
    return Shoup{base[idx], 0};
#else
    if (N1_EXP_NOTW) return Shoup{__builtin_bit_cast(u64, (double)(idx | 1u)), 0};
    return Shoup{((const __attribute__((address_space(1))) u64 *)base)[idx], 0};
#endif
}
__device__ __forceinline__ void ld_w2(const u64 *base, unsigned idx, Shoup (&t)[2]) {
#ifdef TROYHIP_CPU_EMUL
    t[0] = Shoup{base[idx], 0}; t[1] = Shoup{base[idx + 1], 0};
#else
    if (N1_EXP_NOTW) { t[0] = ld_w1(base, idx); t[1] = ld_w1(base, idx + 1); return; }
    const troy_v2ull v = *(const __attribute__((address_space(1))) troy_v2ull *)((const __attribute__((address_space(1))) u64 *)base + idx);
    t[0] = Shoup{v.x, 0}; t[1] = Shoup{v.y, 0};
#endif
}
__device__ __forceinline__ void ld_w4(const u64 *base, unsigned idx, Shoup (&t)[4]) {
#ifdef TROYHIP_CPU_EMUL
    for (int i = 0; i < 4; i++) t[i] = Shoup{base[idx + i], 0};
#else
    if (N1_EXP_NOTW) { for (int i = 0; i < 4; i++) t[i] = ld_w1(base, idx + i); return; }
    const __attribute__((address_space(1))) troy_v2ull *q = (const __attribute__((address_space(1))) troy_v2ull *)((const __attribute__((address_space(1))) u64 *)base + idx);
    const troy_v2ull a = q[0], b = q[1];
    t[0] = Shoup{a.x, 0}; t[1] = Shoup{a.y, 0}; t[2] = Shoup{b.x, 0}; t[3] = Shoup{b.y, 0};
#endif
}

// keeps the LDS / table address arithmetic of a sub-block inside the row loop: hoisted out of it, the ~40 lane-dependent
// addresses of the three rounds would have to live (and be spilled) across the whole kernel; recomputing them costs a few XORs
__device__ __forceinline__ unsigned opaque(unsigned v) {
#ifndef TROYHIP_CPU_EMUL
    asm volatile("" : "+v"(v));
#endif
    return v;
}
#ifdef TROYHIP_CPU_EMUL
#define N1_SCHED_FENCE()
#define N1_PIN_LOADS()
__device__ __forceinline__ void order_after(u64 &, const u64 &) {}
#else
#define N1_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// loads written above this line are ISSUED above it: without it the compiler sinks a prefetch to its first use (there is no barrier or LDS operation
// between the end of a row and the top of the next), i.e. issues it where it is needed
#define N1_PIN_LOADS() asm volatile("" ::: "memory")
// `later` is not touched before `earlier` has its final value (a data dependency the optimiser cannot see through): orders pure arithmetic, which
// scheduling fences do not
__device__ __forceinline__ void order_after(u64 &later, const u64 &earlier) { asm volatile("" : "+v"(later) : "v"(earlier)); }
#endif

__device__ __forceinline__ unsigned uniform_u32(unsigned v) {
#ifdef TROYHIP_CPU_EMUL
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}
// data access with a wave-uniform base and a per-lane 32-bit element offset (global_load/store ... saddr form: no 64-bit address VGPRs)
#ifdef TROYHIP_CPU_EMUL
__device__ __forceinline__ u64 ld_g(const u64 *base, unsigned off) { return base[off]; }
__device__ __forceinline__ u64 ld_g_fwd(const u64 *base, unsigned off) { return base[off]; }
__device__ __forceinline__ void st_g(u64 *base, unsigned off, u64 v) { base[off] = v; }
template <bool NT> __device__ __forceinline__ void st_g_inv(u64 *base, unsigned off, u64 v) { base[off] = v; }
__device__ __forceinline__ ulonglong2 ld_g2(const u64 *base, unsigned off) { return *reinterpret_cast<const ulonglong2 *>(base + off); }
__device__ __forceinline__ void st_g2(u64 *base, unsigned off, ulonglong2 v) { *reinterpret_cast<ulonglong2 *>(base + off) = v; }
#else
#ifndef N1_NT
#define N1_NT 0 // probe: non-temporal row accesses (bit 0 stores, bit 1 every row load, bit 2 the forward kernels' row loads only).  Same-box A/B
#endif          // (profiles/r05_inv_probes.txt): standalone, stores -1.5 % on the integer inverse and loads -2 % on the forward kernel (+2 % on the inverse);
                // inside the steps the consumers of those rows lose what L2 / MALL held for them: headline +0.4 %, 49-bit twin -0.8 %, CKKS chain +-0.5 %.  Off.
__device__ __forceinline__ u64 ld_g(const u64 *base, unsigned off) {
    if (N1_NT & 2) return __builtin_nontemporal_load(((const __attribute__((address_space(1))) u64 *)base) + off);
    return ((const __attribute__((address_space(1))) u64 *)base)[off];
}
__device__ __forceinline__ u64 ld_g_fwd(const u64 *base, unsigned off) { // the forward kernels' row loads
    if (N1_NT & 6) return __builtin_nontemporal_load(((const __attribute__((address_space(1))) u64 *)base) + off);
    return ((const __attribute__((address_space(1))) u64 *)base)[off];
}
__device__ __forceinline__ void st_g(u64 *base, unsigned off, u64 v) {
    if (N1_NT & 1) { __builtin_nontemporal_store(v, ((__attribute__((address_space(1))) u64 *)base) + off); return; }
    ((__attribute__((address_space(1))) u64 *)base)[off] = v;
}
// the stores of the N = 2^15 inverse kernels; NT: streamed past L2's replacement order (N1_NT_INV below)
template <bool NT> __device__ __forceinline__ void st_g_inv(u64 *base, unsigned off, u64 v) {
    if (NT || (N1_NT & 1)) { __builtin_nontemporal_store(v, ((__attribute__((address_space(1))) u64 *)base) + off); return; }
    ((__attribute__((address_space(1))) u64 *)base)[off] = v;
}
__device__ __forceinline__ ulonglong2 ld_g2(const u64 *base, unsigned off) { // off even: 16 bytes
    const troy_v2ull v = (N1_NT & 2) ? __builtin_nontemporal_load((const __attribute__((address_space(1))) troy_v2ull *)(((const __attribute__((address_space(1))) u64 *)base) + off))
                                     : *(const __attribute__((address_space(1))) troy_v2ull *)(((const __attribute__((address_space(1))) u64 *)base) + off);
    ulonglong2 r;
    r.x = v.x;
    r.y = v.y;
    return r;
}
__device__ __forceinline__ void st_g2(u64 *base, unsigned off, ulonglong2 v) {
    troy_v2ull w;
    w.x = v.x;
    w.y = v.y;
    if (N1_NT & 1) { __builtin_nontemporal_store(w, (__attribute__((address_space(1))) troy_v2ull *)(((__attribute__((address_space(1))) u64 *)base) + off)); return; }
    *(__attribute__((address_space(1))) troy_v2ull *)(((__attribute__((address_space(1))) u64 *)base) + off) = w;
}
#endif

// ---- R forward stages on G groups of 2^R register-resident values; tw(st, g, blk) = twiddle of block blk of group g at stage st
template <int G, int R, bool LEAN, bool UNI, class TW> __device__ __forceinline__ void fwd_stages(u64 (&y)[G << R], const TW &tw, const PrimeConst &pc) {
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int half = (1 << R) >> (st + 1);
#pragma unroll
        for (int c = 0; c < (G << (R - 1)) / 4; c++) {
            u64 X[4], Y[4];
            Shoup w[4];
            int ix[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int b = 4 * c + i, g = b >> (R - 1), r = b & ((1 << (R - 1)) - 1);
                const int blk = r / half, k = r % half;
                ix[i] = (g << R) + blk * 2 * half + k;
                X[i] = y[ix[i]];
                Y[i] = y[ix[i] + half];
                w[i] = tw(st, g, blk);
            }
            if (LEAN) ct_bfly4_ng<UNI>(X, Y, w, pc); else ct_bfly4<UNI>(X, Y, w, pc);
#pragma unroll
            for (int i = 0; i < 4; i++) { y[ix[i]] = X[i]; y[ix[i] + half] = Y[i]; }
            N1_SCHED_FENCE(); // one group of four butterflies at a time: interleaving several only multiplies the temporaries (128-VGPR budget)
        }
    }
}
// ---- R inverse stages (gaps grow: 1, 2, ..); LAST: the final stage of the transform (N^-1 folded in, tw gives the scaled twiddle)
// B0 = 0: guarded butterflies, values in [0,4p) throughout.  B0 > 0 ("lean", primes in [2^33, 2^58)): the inputs are below B0 p and
// nothing is subtracted conditionally: a sum output doubles the bound per stage, a difference output leaves its multiplication in
// [0,3p); a stage accepts inputs up to 32p (X + kp - Y with kp = the bound must stay below 64p < 2^64), so the bound may reach 64p at
// the END of a round, where the caller brings it back with lite_reduce4 -- or, in the last round, where the N^-1 multiplication does.
template <int G, int R, bool LAST, bool UNI, int B0 = 0, class TW> __device__ __forceinline__ void inv_stages(u64 (&y)[G << R], const TW &tw, const Shoup inv_n, const PrimeConst &pc) {
    int bound = B0;
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int dist = 1 << st;
        const bool halve = B0 && 2 * bound > 32 && st != R - 1; // the next stage could not take 2 * bound
#pragma unroll
        for (int c = 0; c < (G << (R - 1)) / 4; c++) {
            u64 X[4], Y[4];
            Shoup w[4];
            int ix[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int b = 4 * c + i, g = b >> (R - 1), r = b & ((1 << (R - 1)) - 1);
                const int blk = r / dist, k = r % dist;
                ix[i] = (g << R) + blk * 2 * dist + k;
                X[i] = y[ix[i]];
                Y[i] = y[ix[i] + dist];
                w[i] = tw(st, g, blk);
            }
            if (B0) {
                const u64 kp = pc.p * (u64)bound;
                if (LAST && st == R - 1) gs_bfly4_last_ng<UNI>(X, Y, w, inv_n, kp, pc);
                else gs_bfly4_ng<UNI>(X, Y, w, kp, pc);
                if (halve) csub4(X, kp);
            } else if (LAST && st == R - 1) gs_bfly4_last<UNI>(X, Y, w, inv_n, pc);
            else gs_bfly4<UNI>(X, Y, w, pc);
#pragma unroll
            for (int i = 0; i < 4; i++) { y[ix[i]] = X[i]; y[ix[i] + dist] = Y[i]; }
            N1_SCHED_FENCE();
        }
        if (!halve) bound = 2 * bound > 3 ? 2 * bound : 3; // a difference output is below 3p whatever came in
    }
}

// ---- the same R inverse stages for lean primes ([2^33, 2^58): 64p < 2^64) with a bound tracked PER REGISTER at compile time (round 5).  A sum output
// carries bound(X) + bound(Y), a difference output leaves its multiplication below 3p -- so after a few stages only the registers that were a sum
// several times in a row are large: 1 in 2^k after k stages.  inv_stages (above) tracks ONE bound per stage and therefore reduces every value before a
// round (lite_reduce4 on all 16 / 32 registers) and halves every sum of a critical stage (csub4); here a register is reduced (lite_reduce1: any 64-bit
// value -> below 4p, 4 instructions) only when ITS butterfly could leave 64 bits (bound(X) + bound(Y) > 64), and at the end of the round only the
// registers above EXIT, which is the entry bound E of the next round (its values arrive through LDS from other lanes: every register may hold the
// largest).  Per thread-row of the N = 2^15 transform: 26 reductions instead of 48 + 32 conditional subtractions (bounds: tools/inv_bounds.py).
// All loops unroll completely, so bd[] folds to constants and the tests below cost nothing at run time.
// EXACT (with LAST): exact quotients in the last stage, outputs below 2p.
template <int G, int R, bool LAST, bool UNI, int E, int EXIT, bool EXACT = false, class TW>
__device__ __forceinline__ void inv_stages_lean(u64 (&y)[G << R], const TW &tw, const Shoup inv_n, const u32 mu, const PrimeConst &pc) {
    int bd[G << R];
#pragma unroll
    for (int i = 0; i < (G << R); i++) bd[i] = E;
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int dist = 1 << st;
#pragma unroll
        for (int c = 0; c < (G << (R - 1)) / 4; c++) {
            u64 X[4], Y[4], kp[4];
            Shoup w[4];
            int ix[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int b = 4 * c + i, g = b >> (R - 1), r = b & ((1 << (R - 1)) - 1);
                const int blk = r / dist, k = r % dist;
                ix[i] = (g << R) + blk * 2 * dist + k;
                if (bd[ix[i]] + bd[ix[i] + dist] > 64) { // X + Y and X + kp - Y must stay below 2^64
                    if (bd[ix[i]] > 4) { lite_reduce1(y[ix[i]], mu, pc); bd[ix[i]] = 4; }
                    if (bd[ix[i] + dist] > 4) { lite_reduce1(y[ix[i] + dist], mu, pc); bd[ix[i] + dist] = 4; }
                }
                X[i] = y[ix[i]];
                Y[i] = y[ix[i] + dist];
                kp[i] = pc.p * (u64)bd[ix[i] + dist];
                w[i] = tw(st, g, blk);
            }
            if (LAST && st == R - 1) gs_bfly4_last_ng_k<UNI, EXACT>(X, Y, w, inv_n, kp, pc);
            else gs_bfly4_ng_k<UNI>(X, Y, w, kp, pc);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                y[ix[i]] = X[i];
                y[ix[i] + dist] = Y[i];
                bd[ix[i]] = (LAST && st == R - 1) ? 3 : bd[ix[i]] + bd[ix[i] + dist];
                bd[ix[i] + dist] = 3;
            }
            N1_SCHED_FENCE();
        }
    }
#pragma unroll
    for (int i = 0; i < (G << R); i++)
        if (bd[i] > EXIT) lite_reduce1(y[i], mu, pc);
}

// ---- FP64 forms (fpmod.h; primes in [2^33, 2^50)): y holds the bit patterns of doubles (exact integers, signed lazy range), tw the pairs
// (w, w / p) of PrimeDesc::root_fp / iroot_fp.  Eight instructions per butterfly; the bound walk (where to reduce) is the host's.
template <int G, int R, bool PINV = false, int FENCE = 4, class TW> __device__ __forceinline__ void fp_fwd_stages(u64 (&y)[G << R], const TW &tw, const FpPrime &fc) {
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int half = (1 << R) >> (st + 1);
#pragma unroll
        for (int b = 0; b < (G << (R - 1)); b++) {
            const int g = b >> (R - 1), r = b & ((1 << (R - 1)) - 1);
            const int blk = r / half, k = r % half;
            const int ix = (g << R) + blk * 2 * half + k;
            const Shoup w = tw(st, g, blk);
            const double X = fp_of_bits(y[ix]);
            const double v = PINV ? fp_mulmod_pinv(fp_of_bits(y[ix + half]), fp_of_bits(w.op), fc) : fp_mulmod_wp(fp_of_bits(y[ix + half]), fp_of_bits(w.op), fp_of_bits(w.quo), fc);
            y[ix] = fp_bits(X + v);
            y[ix + half] = fp_bits(X - v);
            if ((b & (FENCE - 1)) == FENCE - 1) N1_SCHED_FENCE(); // FENCE butterflies at a time (four in the integer form): more in flight only multiplies the temporaries
        }
    }
}
// ST0 / ST1: the stages [ST0, ST1) of the R-stage round (the last stage of the transform is run on its own so that a reduction fits in front of it)
template <int G, int R, bool LAST, int ST0, int ST1, bool PINV = false, class TW> __device__ __forceinline__ void fp_inv_stages(u64 (&y)[G << R], const TW &tw, const Shoup inv_n, const FpPrime &fc) {
#pragma unroll
    for (int st = ST0; st < ST1; st++) {
        const int dist = 1 << st;
#pragma unroll
        for (int b = 0; b < (G << (R - 1)); b++) {
            const int g = b >> (R - 1), r = b & ((1 << (R - 1)) - 1);
            const int blk = r / dist, k = r % dist;
            const int ix = (g << R) + blk * 2 * dist + k;
            const Shoup w = tw(st, g, blk);
            const double X = fp_of_bits(y[ix]), Y = fp_of_bits(y[ix + dist]);
            const double sum = X + Y, dif = X - Y;
            y[ix] = fp_bits(LAST && st == R - 1 ? fp_mulmod_wp(sum, fp_of_bits(inv_n.op), fp_of_bits(inv_n.quo), fc) : sum);
            y[ix + dist] = fp_bits(PINV ? fp_mulmod_pinv(dif, fp_of_bits(w.op), fc) : fp_mulmod_wp(dif, fp_of_bits(w.op), fp_of_bits(w.quo), fc));
            if ((b & 3) == 3) N1_SCHED_FENCE();
        }
    }
}
template <int NV> __device__ __forceinline__ void fp_reduce_all(u64 (&y)[NV], const FpPrime &fc) {
#pragma unroll
    for (int i = 0; i < NV; i++) y[i] = fp_bits(fp_reduce(fp_of_bits(y[i]), fc));
}

// the last 10 forward stages (5..14) of sub-block sb (coefficients 1024 sb .. 1024 sb + 1023), in place in the wave's region,
// result canonical to `out` (the sub-block's 1024 coefficients in HBM)
// LOGN: the transform's size; the sub-block rounds are its stages LOGN - 10 .. LOGN - 1 (N = 2^15: 5 .. 14 as the comments say; the smaller sizes of
// ntt1s_*_body shift every stage number down, the code is the same: a sub-block is 1024 coefficients at every size)
// LDS addresses: one per-lane byte offset per round, a literal XOR per access (sw1 and sw2 are linear over GF(2): see inv_subblock)
template <int LOGN, bool LEAN, bool CR, bool FP> __device__ __forceinline__ void fwd_subblock(u64 *lds_base, const unsigned region_words, const unsigned sb, const unsigned lane_in, const PrimeDesc &pd, const PrimeConst &pc, const Mod &m,
                                                                  u64 *out, const Ntt1Args &a, const unsigned mm, const unsigned m_begin, const int stamp0, const bool hf_last,
                                                                  const FpPrime &fc, const u64 *cr_in = nullptr, const Shoup cr_inv = Shoup{0, 0},
                                                                  const u64 *cr_acc = nullptr) {
    (void)a; (void)mm; (void)m_begin; (void)stamp0; (void)hf_last; (void)fc;
    constexpr unsigned A = 1u << (LOGN - 10); // blocks of the first sub-block stage per sub-block row: root index of stage s = 2^s + block
    const unsigned lane = opaque(lane_in);
    const unsigned rb = 8 * region_words;
    N1_PRIO(3);
    {   // round B: stages 5..8 on 16 values, registers = j9..j6, lane = j5..j0; twiddles depend on (sb, register) only: scalar loads
        u64 y[16];
        const unsigned wb = rb + 8 * sw1(lane);
#pragma unroll
        for (int r = 0; r < 16; r++) y[r] = *lds_at(lds_base, wb ^ (8 * sw1(64 * r)));
        if constexpr (FP) {
            if (a.fp_red_mask & 2u) fp_reduce_all<16>(y, fc);
            fp_fwd_stages<1, 4, false, N1_FP_FENCE_B>(y, [&](int st, int, int blk) { return ld_tw_uniform(pd.root + (A << st) + (sb << st) + blk); }, fc);
        } else
        fwd_stages<1, 4, LEAN, true>(y, [&](int st, int, int blk) { return ld_tw_uniform(pd.root + (A << st) + (sb << st) + blk); }, pc);
#pragma unroll
        for (int r = 0; r < 16; r++) *lds_at(lds_base, wb ^ (8 * sw1(64 * r))) = y[r];
    }
    TROY_WAVE_SYNC();
    N1_STAMP(stamp0);
    N1_SCHED_FENCE(); // the twiddle loads below stay below: hoisted over round B they would not fit the register budget (the FP64 instances, whose seven
                      // twiddles are single doubles, were tried with them in front of round B at N <= 2^14, where they fit: no measurable change)
    N1_PRIO(2);
    {   // round C1: stages 9..11 on 8 values, registers = j5 j4 j3, lane = (j9..j6, j1 j0), iteration = j2; both iterations share
        // the seven twiddles
        const unsigned h = lane >> 2, low = lane & 3;
        const unsigned b9 = 16 * sb + h;
        const Shoup t9 = FP ? ld_w1(pd.root_w, 16 * A + b9) : ld_tw(pd.root, 16 * A + b9);
        Shoup t10[2], t11[4];
        if constexpr (FP) {
            ld_w2(pd.root_w, 32 * A + 2 * b9, t10);
            ld_w4(pd.root_w, 64 * A + 4 * b9, t11);
        } else {
#pragma unroll
        for (int i = 0; i < 2; i++) t10[i] = ld_tw(pd.root, 32 * A + 2 * b9 + i);
#pragma unroll
        for (int i = 0; i < 4; i++) t11[i] = ld_tw(pd.root, 64 * A + 4 * b9 + i);
        }
        const unsigned wc0 = rb + 8 * sw1(64 * h + low);
#pragma unroll 1
        for (unsigned it = 0; it < 2; it++) {
            u64 y[8];
            const unsigned wc = it ? wc0 ^ (8 * sw1(4)) : wc0;
#pragma unroll
            for (int r = 0; r < 8; r++) y[r] = *lds_at(lds_base, wc ^ (8 * sw1(8 * r)));
            if constexpr (FP) {
                if (a.fp_red_mask & 4u) fp_reduce_all<8>(y, fc);
                fp_fwd_stages<1, 3, true>(y, [&](int st, int, int blk) { return st == 0 ? t9 : (st == 1 ? t10[blk] : t11[blk]); }, fc);
            } else
            fwd_stages<1, 3, LEAN, false>(y, [&](int st, int, int blk) { return st == 0 ? t9 : (st == 1 ? t10[blk] : t11[blk]); }, pc);
#pragma unroll
            for (int r = 0; r < 8; r++) *lds_at(lds_base, wc ^ (8 * sw1(8 * r))) = y[r];
        }
    }
    TROY_WAVE_SYNC();
    N1_STAMP(stamp0 + 1);
    N1_SCHED_FENCE();
    if (hf_last) { N1_PRIO(2); } else { N1_PRIO(1); }
    // round C2: stages 12..14 on 8 consecutive coefficients, u = j9..j3 = lane + 64 iteration
#pragma unroll 1
    for (unsigned it = 0; it < 2; it++) {
        const unsigned u = lane + 64 * it;
        u64 y[8];
        const unsigned b12 = 128 * sb + u;
        const Shoup t12 = FP ? ld_w1(pd.root_w, 128 * A + b12) : ld_tw(pd.root, 128 * A + b12);
        Shoup t13[2], t14[4];
        if constexpr (FP) {
            ld_w2(pd.root_w, 256 * A + 2 * b12, t13);
            ld_w4(pd.root_w, 512 * A + 4 * b12, t14);
        } else {
#pragma unroll
        for (int i = 0; i < 2; i++) t13[i] = ld_tw(pd.root, 256 * A + 2 * b12 + i);
#pragma unroll
        for (int i = 0; i < 4; i++) t14[i] = ld_tw(pd.root, 512 * A + 4 * b12 + i);
        }
        const unsigned wr2 = rb + 8 * sw1(8 * lane) ^ (it ? 8 * sw1(512) : 0u), ww2 = rb + 8 * wb_swz(8 * lane) ^ (it ? 8 * wb_swz(512) : 0u);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(lds_at(lds_base, wr2 ^ (8 * sw1(2 * q))));
            y[2 * q] = v.x;
            y[2 * q + 1] = v.y;
        }
        if constexpr (FP) {
            if (a.fp_red_mask & 8u) fp_reduce_all<8>(y, fc);
            fp_fwd_stages<1, 3, true>(y, [&](int st, int, int blk) { return st == 0 ? t12 : (st == 1 ? t13[blk] : t14[blk]); }, fc);
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = fp_canonical(fp_of_bits(y[i]), fc, m.p);
        } else {
        fwd_stages<1, 3, LEAN, false>(y, [&](int st, int, int blk) { return st == 0 ? t12 : (st == 1 ? t13[blk] : t14[blk]); }, pc);
        }
        if (FP) {
        } else if (LEAN) {
            lean_final<8>(y, make_lean_final(m.p, m.cr1), pc);
        } else {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                u64 v[4] = {y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]};
                reduce4_from_8p(v, pc);
#pragma unroll
                for (int i = 0; i < 4; i++) y[4 * q + i] = v[i];
            }
        }
        if (CR && !(N1_FWD_STORE_VIA_LDS && N1_CR_LINEAR)) { // y = NTT(corr), canonical: out = (in + p - y) * inv mod p, stored or added to what is there (Ntt1Corr)
            const Shoup iq[4] = {cr_inv, cr_inv, cr_inv, cr_inv};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                u64 w[4], q[4], r[4];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const ulonglong2 v = ld_g2(cr_in, 8 * u + 4 * h + 2 * e);
                    w[2 * e] = v.x + pc.p - y[4 * h + 2 * e];
                    w[2 * e + 1] = v.y + pc.p - y[4 * h + 2 * e + 1];
                }
                mulhi_approx4_u(q, w, iq);
#pragma unroll
                for (int i = 0; i < 4; i++) r[i] = mul_acc0_u(w[i], cr_inv.op, q[i], pc.negp); // [0, 3p)
                if (cr_acc) { // what the result is added to: the output itself, or the base polynomial of a rotation (null: nothing, start from zero)
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const ulonglong2 v = ld_g2(cr_acc, 8 * u + 4 * h + 2 * e);
                        r[2 * e] += v.x;
                        r[2 * e + 1] += v.y;
                    }
                }
                csub4(r, pc.two_p);
                csub4(r, pc.p);
#pragma unroll
                for (int i = 0; i < 4; i++) y[4 * h + i] = r[i];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            ulonglong2 v;
            v.x = y[2 * q];
            v.y = y[2 * q + 1];
            if (N1_FWD_STORE_VIA_LDS) *reinterpret_cast<ulonglong2 *>(lds_at(lds_base, ww2 ^ (8 * wb_swz(2 * q)))) = v; // into the eight slots it was read from (sw2: these 16-byte writes are 2-way conflicted under sw1)
            else st_g2(out, 8 * u + 2 * q, v);
        }
    }
    if (N1_FWD_STORE_VIA_LDS) {
        // A thread ends with eight consecutive coefficients: stored from there, an instruction writes 16 bytes per lane every 64 bytes -- 64 partial
        // lines, four instructions per line.  Through the wave's region (its content is dead now) the same data leaves lane-linearly: eight
        // instructions of one contiguous KiB each.  (The inverse kernel's loads and stores are lane-linear by construction; this was the forward
        // kernel's one scattered access.)
        TROY_WAVE_SYNC();
        if constexpr (CR && N1_CR_LINEAR) {
            // the divide-and-round epilogue on the way out, in the same layout: y = NTT(corr), canonical; out = (in + p - y) * inv mod p, stored or added to
            // what is there (Ntt1Corr) -- `in` and the accumulation target are read a contiguous KiB per instruction like the stores
            const Shoup iq[4] = {cr_inv, cr_inv, cr_inv, cr_inv};
            u64 *const R = lds_base + region_words;
#pragma unroll 1
            for (unsigned i = 0; i < 8; i += 2) {
                const unsigned j0 = 128 * i + 2 * lane, j1 = j0 + 128;
                const ulonglong2 ya = *reinterpret_cast<const ulonglong2 *>(R + wb_swz(j0)), yb = *reinterpret_cast<const ulonglong2 *>(R + wb_swz(j1));
                const ulonglong2 ia = ld_g2(cr_in, j0), ib = ld_g2(cr_in, j1);
                u64 w[4] = {ia.x + pc.p - ya.x, ia.y + pc.p - ya.y, ib.x + pc.p - yb.x, ib.y + pc.p - yb.y}, q[4], r[4];
                mulhi_approx4_u(q, w, iq);
#pragma unroll
                for (int e = 0; e < 4; e++) r[e] = mul_acc0_u(w[e], cr_inv.op, q[e], pc.negp); // [0, 3p)
                if (cr_acc) { // what the result is added to: the output itself, or the base polynomial of a rotation (null: nothing, start from zero)
                    const ulonglong2 ca = ld_g2(cr_acc, j0), cb = ld_g2(cr_acc, j1);
                    r[0] += ca.x; r[1] += ca.y; r[2] += cb.x; r[3] += cb.y;
                }
                csub4(r, pc.two_p);
                csub4(r, pc.p);
                st_g2(out, j0, ulonglong2{r[0], r[1]});
                st_g2(out, j1, ulonglong2{r[2], r[3]});
            }
        } else {
        u64 *const R = lds_base + region_words;
#pragma unroll 1
        for (unsigned i = 0; i < 8; i += N1_FWD_STORE_UNROLL) { // a few at a time: the waiting half of the row and the next row's prefetch hold most of the registers
#pragma unroll
            for (unsigned k = 0; k < N1_FWD_STORE_UNROLL; k++) {
                const unsigned j = 128 * (i + k) + 2 * lane;
                st_g2(out, j, *reinterpret_cast<const ulonglong2 *>(R + wb_swz(j)));
            }
        }
        }
    }
}

template <bool LEAN, bool CR, bool FP> __device__ __forceinline__ void ntt1_fwd_body(const Ntt1Args &a) {
    // 16 regions of 8 KiB + 32 KiB in which four of the sixteen registers of the waiting half are parked (the other twelve stay in
    // VGPRs): all 160 KiB of the CU
    __shared__ __attribute__((aligned(16))) u64 lds[20 * 1024];
    const unsigned tid = threadIdx.x, lane = tid & 63;
#ifdef TROYHIP_CPU_EMUL
    const unsigned wv = tid >> 6;
#else
    const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    // the byte tables of the argument block are read with vector loads: tell the compiler the results are wave-uniform
    unsigned sidx, chunk;
    if (!n1_unit(a, sidx, chunk)) return;
    const unsigned slot = uniform_u32(a.slots[sidx]);
    PrimeDesc pd = a.primes[uniform_u32(a.map.id[slot])];
    if constexpr (FP) pd.root = pd.root_fp;
    const FpPrime fc = make_fp_prime_uniform(FP ? pd.p : 1);
    const PrimeConst pc = make_prime_const(pd.p);
    const Mod m = mod_of(pd);
    const unsigned m_begin = chunk * a.rows_per_wg;
    const unsigned m_end = (m_begin + a.rows_per_wg < a.m_total) ? m_begin + a.rows_per_wg : a.m_total;
    const unsigned inner = a.map.inner, period = a.map.period;
    auto row_of = [&](unsigned mm) -> u64 {
        const unsigned o = mm / inner, k = mm - o * inner;
        return (((u64)o * period + slot) * inner + k) << N1_LOGN;
    };
    const u64 *in_base = a.src ? a.src : a.data;
    // CR (inner == 1): row mm of this slot is built from the single source row cr_last[mm] and leaves through the epilogue
    auto in_row = [&](unsigned mm) -> const u64 * { return CR ? a.cr_last + ((u64)mm << N1_LOGN) : in_base + row_of(mm); };
    const Shoup cr_inv = CR ? Shoup{((const u64 *)(a.cr_inv + slot))[0], ((const u64 *)(a.cr_inv + slot))[1]} : Shoup{0, 0};
    const u64 cr_add = CR ? pd.p - barrett64(a.cr_half, m) : 0; // p - [half]_p
    u64 *const region = lds + 1024 * wv;
    n1_stagger();
    // registers of round A: xe[r'] = coefficient 1024 (2 r') + tid, xo[r'] = coefficient 1024 (2 r' + 1) + tid
    u64 xe[16], xo[16];
    // loads and stores of round A: ONE wave-uniform base (the limb) plus a 32-bit per-thread offset (one v_add per access; 32
    // uniform bases would not fit the scalar registers next to the twiddles)
    auto load_half = [&](u64 (&x)[16], const u64 *rowp, unsigned odd) {
        const unsigned t = opaque(tid); // the 16 offsets are formed here, next to the loads, not carried through the kernel
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = ld_g_fwd(rowp, t + 2048 * r + 1024 * odd);
    };
    load_half(xe, in_row(m_begin), 0);
    load_half(xo, in_row(m_begin), 1);
    u64 *const wlo = lds + sw1(tid), *const whi = wlo + 8 * 1024; // LDS addresses of this thread's element in regions 0..7 / 8..15
    u64 *const park = lds + 16 * 1024 + tid;
    for (unsigned mm = m_begin; mm < m_end; mm++) {
        u64 *const out = CR ? a.cr_out + (u64)(mm / a.cr_group) * a.cr_out_gstride + (u64)(mm % a.cr_group) * a.cr_out_ostride + ((u64)slot << N1_LOGN) : a.data + row_of(mm);
        // accumulate: onto out, or -- rotations -- onto (base, 0): member 0 of a group reads the base polynomial, the others start from zero
        const u64 *const cacc = !CR || !a.cr_accumulate ? nullptr
                                : !a.cr_base            ? out
                                : ((int)(mm % a.cr_group) < a.cr_base_polys) ? a.cr_base + (u64)(mm / a.cr_group) * a.cr_base_gstride + (u64)(mm % a.cr_group) * a.cr_out_ostride + ((u64)slot << N1_LOGN)
                                                         : nullptr;
        const u64 *const cin = CR ? a.cr_in + (a.cr_in_gstride ? (u64)(mm / a.cr_group) * a.cr_in_gstride + (u64)(mm % a.cr_group) * a.cr_in_ostride : (u64)mm * a.cr_in_ostride) + ((u64)slot << N1_LOGN) : nullptr;
        // what the loaded words become before the first butterfly, one half at a time: the odd half was requested LAST (one row ahead of its use by
        // a few hundred instructions only), so in the FP64 kernels -- where this step is the first use of the words -- it is prepared after the even
        // half's four stages, whose ~520 instructions cover its HBM latency (prepared up front it cost 7 % of the row: s_waitcnt vmcnt(0) at the loop top)
        auto prepare = [&](u64 (&xx)[16]) {
            if (CR) { // corr = [(last + half) mod qx]_p + (p - [half]_p), the residue lazily below 4p: below 5p, inside what the butterflies take
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    u64 v[4] = {xx[4 * g] + a.cr_half, xx[4 * g + 1] + a.cr_half, xx[4 * g + 2] + a.cr_half, xx[4 * g + 3] + a.cr_half};
                    csub4(v, a.cr_qx);
                    lite_reduce4(v, (u32)pd.cr1, pc);
#pragma unroll
                    for (int i = 0; i < 4; i++) xx[4 * g + i] = FP ? fp_bits(fp_from_u64(v[i]) + (double)cr_add) : v[i] + cr_add; // FP64: 4p < 2^52 converts exactly
                }
            } else if constexpr (FP) { // canonical residues become doubles (exact below 2^52)
#pragma unroll
                for (int r = 0; r < 16; r++) xx[r] = fp_bits(fp_from_u64(xx[r]));
            }
            if constexpr (FP) {
                if (a.fp_red_mask & 1u) fp_reduce_all<16>(xx, fc);
            }
        };
        prepare(xe);
        if constexpr (!FP || N1_FP_ODD_EARLY) prepare(xo);
        N1_STAMP(0);
        // round A: stages 0..3 on the even and on the odd registers (the odd ones were requested last), then stage 4 across
        auto twA = [&](int st, int, int blk) { return ld_tw_uniform((pd.root + (1u << st) + blk)); };
        N1_PRIO(1);
        if constexpr (FP) {
            fp_fwd_stages<1, 4>(xe, twA, fc);
            if constexpr (!N1_FP_ODD_EARLY) {
#pragma unroll
                for (int r = 0; r < 16; r++) order_after(xo[r], xe[r]);
                prepare(xo);
            }
            fp_fwd_stages<1, 4>(xo, twA, fc);
#pragma unroll
            for (int r = 0; r < 16; r++) { // stage 4: (xe[r], xo[r]) with the twiddle of block r
                const Shoup w = ld_tw_uniform((pd.root + 16 + r));
                const double X = fp_of_bits(xe[r]);
                const double v = fp_mulmod_wp(fp_of_bits(xo[r]), fp_of_bits(w.op), fp_of_bits(w.quo), fc);
                xe[r] = fp_bits(X + v);
                xo[r] = fp_bits(X - v);
            }
        } else {
        fwd_stages<1, 4, LEAN, true>(xe, twA, pc);
        N1_PRIO(0);
        fwd_stages<1, 4, LEAN, true>(xo, twA, pc);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            u64 X[4], Y[4];
            Shoup w[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { X[i] = xe[4 * c + i]; Y[i] = xo[4 * c + i]; w[i] = ld_tw_uniform((pd.root + 16 + 4 * c + i)); }
            if (LEAN) ct_bfly4_ng<true>(X, Y, w, pc); else ct_bfly4<true>(X, Y, w, pc);
#pragma unroll
            for (int i = 0; i < 4; i++) { xe[4 * c + i] = X[i]; xo[4 * c + i] = Y[i]; }
            N1_SCHED_FENCE();
        }
        }
        N1_STAMP(1);
        // sub-block 2 r' (2 r' + 1) = xe[r'] (xo[r']) of all threads; half hf = sub-blocks 16 hf .. 16 hf + 15
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            if (hf == 1 || mm != m_begin) __syncthreads(); // every wave is done with the regions' previous content
            N1_STAMP(2 + 6 * hf);
            if (hf == 1) {
#pragma unroll
                for (int r = 0; r < 2; r++) { xe[14 + r] = park[2048 * r]; xo[14 + r] = park[2048 * r + 1024]; }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                wlo[1024 * (2 * r)] = xe[8 * hf + r];
                wlo[1024 * (2 * r + 1)] = xo[8 * hf + r];
                whi[1024 * (2 * r)] = xe[8 * hf + 4 + r];
                whi[1024 * (2 * r + 1)] = xo[8 * hf + 4 + r];
            }
            if (hf == 0) {
#pragma unroll
                for (int r = 0; r < 2; r++) { park[2048 * r] = xe[14 + r]; park[2048 * r + 1024] = xo[14 + r]; }
            }
            __syncthreads();
            N1_STAMP(3 + 6 * hf);
            if (hf == 1) {
                if (mm + 1 < m_end) load_half(xe, in_row(mm + 1), 0); // all 32 registers are free now: request the next limb's even half
                else {
                    // the last row: say that the registers are dead -- otherwise the old values are live across the branch, and the compiler keeps them
                    // alive by SPILLING eight of them in every row (64 B of scratch stores per thread and row for a reload that runs once)
#pragma unroll
                    for (int r = 0; r < 16; r++) xe[r] = 0;
                }
            }
            // (the FP64 instances compute faster than the odd half of the next row arrives; requesting it earlier -- after round B of the second half --
            // was measured twice, rounds 3 and 4: the 32 registers it holds through rounds C1 / C2 cost 36-52 B of scratch and 5 % of the kernel)
            fwd_subblock<N1_LOGN, LEAN, CR, FP>(lds, 1024 * wv, 16 * hf + wv, lane, pd, pc, m, out + 1024 * (16 * hf + wv), a, mm, m_begin, 4 + 6 * hf, hf == 1, fc, CR ? cin + 1024 * (16 * hf + wv) : nullptr,
                                       cr_inv, CR && cacc ? cacc + 1024 * (16 * hf + wv) : nullptr);
            N1_STAMP(7 + 6 * hf);
        }
        if (mm + 1 < m_end) {
            load_half(xo, in_row(mm + 1), 1);
            if constexpr (FP && !N1_FP_ODD_EARLY) N1_PIN_LOADS();
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) xo[r] = 0; // dead after the last row (see xe above)
        }
    }
}
template <bool LEAN, bool CR> __global__ __launch_bounds__(N1_THREADS) void ntt1_fwd_kernel(Ntt1Args a) { ntt1_fwd_body<LEAN, CR, false>(a); }
// the FP64 instances (primes in [2^33, 2^50)): kernels of their own name, as in ntt2.hip
template <bool CR> __global__ __launch_bounds__(N1_THREADS) void ntt1_fwd_fp_kernel(Ntt1Args a) { ntt1_fwd_body<true, CR, true>(a); }

// the first 10 inverse stages (14..5) of sub-block sb: input y (this lane's coefficients 8 u + r, u = lane + 64 i, in y[8 i + r]),
// result left in the wave's region (position sw2(j))
// LDS addresses (round 5): sw2 is linear over GF(2) -- every term is a shifted bit field XORed in -- so for an index j = A + c whose lane-dependent part A and
// compile-time part c occupy disjoint bits, sw2(j) = sw2(A) ^ sw2(c).  Each round forms ONE per-lane byte offset W = region + 8 sw2(A) and every access
// is W ^ (8 sw2(c)) with a literal: one v_xor_b32 per LDS instruction instead of the three-input XOR plus shift-and-add the generic expression compiled to
// (2 -> 1 VALU instructions on each of the ~180 LDS accesses of a thread-row).  The region is 8 KiB-aligned inside the array, so the XOR never reaches its bits.
template <int LOGN, bool LEAN, bool FP> __device__ __forceinline__ void inv_subblock(u64 (&yin)[16], u64 *lds_base, const unsigned region_words, const unsigned sb, const unsigned lane_in, const PrimeDesc &pd, const PrimeConst &pc,
                                                                  const FpPrime &fc, const unsigned fp_mask) {
    const Shoup none{0, 0};
    constexpr unsigned NN = 1u << LOGN; // the inverse table keeps the stage with m blocks at offset N - 2 m + 1 (src/utils/ntt.cpp:49-54)
    const unsigned lane = opaque(lane_in);
    const unsigned rb = 8 * region_words;
    (void)fc; (void)fp_mask;
    if constexpr (FP) { // canonical residues become doubles (round D' never needs a reduction: 8 p < 2^53)
#pragma unroll
        for (int i = 0; i < 16; i++) yin[i] = fp_bits(fp_from_u64(yin[i]));
    }
    // round D': stages 14, 13, 12 on 8 consecutive coefficients
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const unsigned u = lane + 64 * it;
        u64 y[8];
#pragma unroll
        for (int r = 0; r < 8; r++) y[r] = yin[8 * it + r];
        const unsigned b12 = 128 * sb + u;
        Shoup t14[4], t13[2];
        if constexpr (FP) { // iroot_w is iroot shifted by the "+ 1" of every stage offset: the groups are aligned
            ld_w4(pd.iroot_w, 4 * b12, t14);
            ld_w2(pd.iroot_w, NN / 2 + 2 * b12, t13);
        } else {
#pragma unroll
        for (int i = 0; i < 4; i++) t14[i] = ld_tw(pd.iroot, 1 + 4 * b12 + i);
#pragma unroll
        for (int i = 0; i < 2; i++) t13[i] = ld_tw(pd.iroot, NN / 2 + 1 + 2 * b12 + i);
        }
        const Shoup t12 = FP ? ld_w1(pd.iroot_w, NN - NN / 4 + b12) : ld_tw(pd.iroot, NN - NN / 4 + 1 + b12);
        if constexpr (FP) fp_inv_stages<1, 3, false, 0, 3, true>(y, [&](int st, int, int blk) { return st == 0 ? t14[blk] : (st == 1 ? t13[blk] : t12); }, none, fc);
        else if constexpr (LEAN) // inputs below 2p (stored limbs are canonical) -> 16p at most (register 0), no reduction
        inv_stages_lean<1, 3, false, false, 2, 64>(y, [&](int st, int, int blk) { return st == 0 ? t14[blk] : (st == 1 ? t13[blk] : t12); }, none, (u32)pd.cr1, pc);
        else
        inv_stages<1, 3, false, false, 0>(y, [&](int st, int, int blk) { return st == 0 ? t14[blk] : (st == 1 ? t13[blk] : t12); }, none, pc);
        const unsigned wd = rb + 8 * sw2(8 * lane); // j = 8 (lane + 64 it) + 2 q: lane in bits 3..8, the rest in bits 1, 2, 9
#pragma unroll
        for (int q = 0; q < 4; q++) {
            ulonglong2 v;
            v.x = y[2 * q];
            v.y = y[2 * q + 1];
            *reinterpret_cast<ulonglong2 *>(lds_at(lds_base, wd ^ (8 * sw2(512 * it + 2 * q)))) = v;
        }
        N1_SCHED_FENCE();
    }
    TROY_WAVE_SYNC();
    {   // round C': stages 11, 10, 9, registers = j3 j4 j5 (bit 0 = j3)
        const unsigned h = lane >> 2, low = lane & 3;
        const unsigned b9 = 16 * sb + h;
        Shoup t11[4], t10[2];
        if constexpr (FP) {
            ld_w4(pd.iroot_w, NN - NN / 8 + 4 * b9, t11);
            ld_w2(pd.iroot_w, NN - NN / 16 + 2 * b9, t10);
        } else {
#pragma unroll
        for (int i = 0; i < 4; i++) t11[i] = ld_tw(pd.iroot, NN - NN / 8 + 1 + 4 * b9 + i);
#pragma unroll
        for (int i = 0; i < 2; i++) t10[i] = ld_tw(pd.iroot, NN - NN / 16 + 1 + 2 * b9 + i);
        }
        const Shoup t9 = FP ? ld_w1(pd.iroot_w, NN - NN / 32 + b9) : ld_tw(pd.iroot, NN - NN / 32 + 1 + b9);
        const unsigned wc0 = rb + 8 * sw2(64 * h + low); // j = 64 h + 8 r + 4 it + low: lane in bits 0, 1, 6..9, (r, it) in bits 2..5
#pragma unroll 1
        for (unsigned it = 0; it < 2; it++) {
            u64 y[8];
            const unsigned wc = it ? wc0 ^ (8 * sw2(4)) : wc0;
#pragma unroll
            for (int r = 0; r < 8; r++) y[r] = *lds_at(lds_base, wc ^ (8 * sw2(8 * r)));
            if constexpr (FP) {
                if (fp_mask & 2u) fp_reduce_all<8>(y, fc);
                fp_inv_stages<1, 3, false, 0, 3, true>(y, [&](int st, int, int blk) { return st == 0 ? t11[blk] : (st == 1 ? t10[blk] : t9); }, none, fc);
            } else if constexpr (LEAN) // 16p in; registers 0 and 4 are reduced before the third stage, register 1 on the way out: 8p out
            inv_stages_lean<1, 3, false, false, 16, 8>(y, [&](int st, int, int blk) { return st == 0 ? t11[blk] : (st == 1 ? t10[blk] : t9); }, none, (u32)pd.cr1, pc);
            else
            inv_stages<1, 3, false, false, 0>(y, [&](int st, int, int blk) { return st == 0 ? t11[blk] : (st == 1 ? t10[blk] : t9); }, none, pc);
#pragma unroll
            for (int r = 0; r < 8; r++) *lds_at(lds_base, wc ^ (8 * sw2(8 * r))) = y[r];
        }
    }
    TROY_WAVE_SYNC();
    {   // round B': stages 8..5, registers = j6..j9 (bit 0 = j6); twiddles depend on (sb, register) only
        u64 y[16];
        const unsigned wb = rb + 8 * sw2(lane); // j = 64 r + lane: lane in bits 0..5, r in bits 6..9
#pragma unroll
        for (int r = 0; r < 16; r++) y[r] = *lds_at(lds_base, wb ^ (8 * sw2(64 * r)));
        if constexpr (FP) {
            if (fp_mask & 4u) fp_reduce_all<16>(y, fc);
            fp_inv_stages<1, 4, false, 0, 4>(y, [&](int st, int, int blk) { return ld_tw_uniform(pd.iroot + (NN - ((NN >> 6) >> st) + 1) + ((8 * sb) >> st) + blk); }, none, fc);
        } else if constexpr (LEAN) // 8p in, four registers reduced before the third stage, register 1 on the way out: 8p out
        inv_stages_lean<1, 4, false, true, 8, 8>(y, [&](int st, int, int blk) { return ld_tw_uniform(pd.iroot + (NN - ((NN >> 6) >> st) + 1) + ((8 * sb) >> st) + blk); }, none, (u32)pd.cr1, pc);
        else
        inv_stages<1, 4, false, true, 0>(y, [&](int st, int, int blk) { return ld_tw_uniform(pd.iroot + (NN - ((NN >> 6) >> st) + 1) + ((8 * sb) >> st) + blk); }, none, pc);
#pragma unroll
        for (int r = 0; r < 16; r++) *lds_at(lds_base, wb ^ (8 * sw2(64 * r))) = y[r];
    }
}

template <bool LEAN, bool MD, bool FP> __device__ __forceinline__ void ntt1_inv_body(const Ntt1Args &a) {
    __shared__ __attribute__((aligned(16))) u64 lds[16 * 1024];
    const unsigned tid = threadIdx.x, lane = tid & 63;
#ifdef TROYHIP_CPU_EMUL
    const unsigned wv = tid >> 6;
#else
    const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    // the byte tables of the argument block are read with vector loads: tell the compiler the results are wave-uniform
    unsigned sidx, chunk;
    if (!n1_unit(a, sidx, chunk)) return;
    const unsigned slot = uniform_u32(a.slots[sidx]);
    PrimeDesc pd = a.primes[uniform_u32(a.map.id[slot])];
    if constexpr (FP) { pd.iroot = pd.iroot_fp; pd.inv_n = pd.inv_n_fp; pd.iroot_last_scaled = pd.iroot_last_scaled_fp; }
    const FpPrime fc = make_fp_prime_uniform(FP ? pd.p : 1);
    const PrimeConst pc = make_prime_const(pd.p);
    const unsigned m_begin = chunk * a.rows_per_wg;
    const unsigned m_end = (m_begin + a.rows_per_wg < a.m_total) ? m_begin + a.rows_per_wg : a.m_total;
    const unsigned inner = a.map.inner, period = a.map.period;
    auto row_of = [&](unsigned mm) -> u64 {
        const unsigned o = mm / inner, k = mm - o * inner;
        return (((u64)o * period + slot) * inner + k) << N1_LOGN;
    };
    u64 *const region = lds + 1024 * wv;
    // LDS-DMA staging of a sub-block into the wave's region, already in sw1 order: instruction i fills bytes [1024 i, 1024 i + 1024)
    // of the region lane-linearly, so lane l fetches the 16-byte unit that belongs at position 128 i + 2 l (sw1 is an involution)
    auto stage_issue = [&](const u64 *sub) {
        if (N1_INV_EXP & 1) return;
        const unsigned l = opaque(lane);
#pragma unroll
        for (int i = 0; i < 8; i++) TROY_GLDS16_POL(sub + sw2_inv(128 * i + 2 * l), region + 128 * i, N1_DMA_POL);
    };
    auto load16 = [&](u64 (&y)[16], const u64 *sub) { // y[8 i + r] = coefficient 8 (lane + 64 i) + r
        if (N1_INV_EXP & 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) y[i] = (u64)(opaque(lane) + 64 * i) * 0x9E3779B97F4A7C15ull >> 7;
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const ulonglong2 v = ld_g2(sub, 8 * (lane + 64 * i) + 2 * q);
                y[8 * i + 2 * q] = v.x;
                y[8 * i + 2 * q + 1] = v.y;
            }
    };
    // out of place: the first touch of a row reads it from the source (strided batch: a.src_ostride words between outer indices)
    auto in_of = [&](unsigned mm) -> const u64 * {
        if (!a.src) return a.data + row_of(mm);
        const unsigned o = mm / inner, k = mm - o * inner;
        return a.src + (u64)o * a.src_ostride + ((u64)(slot * inner + k) << N1_LOGN);
    };
    n1_stagger();
    stage_issue(in_of(m_begin) + 1024 * wv);
#if N1_INV_TOP == 3
    TROY_WAIT_VMEM(); // the first row's staged half (every later row's is waited for in front of the previous row's stores, below)
#endif
    // two base registers for the 32 exchange reads: the immediate offset of a ds_read reaches 64 KiB, so rlo + 8 KiB r and rhi + 8 KiB r need nothing else.
    // (rhi's base is made opaque: the compiler otherwise folds the 64 KiB into EIGHT per-lane addresses it keeps live across the row loop, and spills)
    const u64 *const rlo = lds + sw2(tid), *const rhi = lds_at(lds, opaque(8 * (8 * 1024 + sw2(tid))));
    for (unsigned mm = m_begin; mm < m_end; mm++) {
        u64 *const row = a.data + row_of(mm);
        u64 x[32]; // x[r] = coefficient tid of sub-block r after its 10 stages
        u64 y[16], y1[16];
#if N1_INV_TOP == 3
        // probe: the staged half was waited for when nothing younger than it was in flight (before the previous row's stores), so no wait here has a fresh
        // store in front of it.  Measured neutral: what the removal probe N1_INV_EXP = 2 attributes to the stores (10 % of the integer kernel, a third of the
        // FP64 one) is their HBM traffic, not a drain at this point.
        load16(y1, in_of(mm) + 1024 * (16 + wv));
#elif N1_INV_TOP == 0 || defined(TROYHIP_CPU_EMUL)
        load16(y1, in_of(mm) + 1024 * (16 + wv)); // second half's input: in flight while the first half is transformed
        TROY_WAIT_VMEM();                   // the staged first half has landed (vmcnt counts in order: this also waits for y1)
#elif N1_INV_TOP == 1 // probe: the second half requested AFTER the wait for the staged first half -- its latency sits under the first half's ten stages
        TROY_WAIT_VMEM();
        load16(y1, in_of(mm) + 1024 * (16 + wv));
#else                 // probe: requested first, but the wait leaves its eight loads in flight (reads return in order: everything older has landed)
        load16(y1, in_of(mm) + 1024 * (16 + wv));
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
        const unsigned ol = opaque(lane);   // recomputed per row: hoisted out of the loop these eight addresses are spilled
        const unsigned ws = 8 * 1024 * wv + 8 * sw2(8 * ol); // as round D' of inv_subblock: one per-lane offset, a literal XOR per access
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(lds_at(lds, ws ^ (8 * sw2(512 * i + 2 * q))));
                y[8 * i + 2 * q] = v.x;
                y[8 * i + 2 * q + 1] = v.y;
            }
        TROY_WAVE_SYNC();
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            inv_subblock<N1_LOGN, LEAN, FP>(hf ? y1 : y, lds, 1024 * wv, 16 * hf + wv, lane, pd, pc, fc, a.fp_red_mask);
            if (N1_INV_EXP & 4) TROY_WAVE_SYNC(); else __syncthreads();
#pragma unroll
            for (int r = 0; r < 8; r++) { x[16 * hf + r] = rlo[1024 * r]; x[16 * hf + 8 + r] = rhi[1024 * r]; }
            if (N1_INV_EXP & 4) TROY_WAVE_SYNC(); else __syncthreads();
        }
        if (mm + 1 < m_end) stage_issue(in_of(mm + 1) + 1024 * wv); // the regions are free during round A'
        // round A': stages 4..0 across the 32 sub-blocks, N^-1 folded into the last one
        auto twA = [&](int st, int, int blk) { return st == 4 ? pd.iroot_last_scaled : ld_tw_uniform((pd.iroot + (N1_N - (32u >> st) + 1) + blk)); };
        if constexpr (FP) {
            if (a.fp_red_mask & 8u) fp_reduce_all<32>(x, fc);
            fp_inv_stages<1, 5, true, 0, 4>(x, twA, pd.inv_n, fc);
            if (a.fp_red_mask & 16u) fp_reduce_all<32>(x, fc);
            fp_inv_stages<1, 5, true, 4, 5>(x, twA, pd.inv_n, fc);
#pragma unroll
            for (int i = 0; i < 32; i++) x[i] = fp_canonical(fp_of_bits(x[i]), fc, pd.p);
        } else if constexpr (LEAN) {
            // 8p in (the sub-block rounds' exit bound); four of the 32 registers are reduced on the way.  Plain stores: exact quotients in the last
            // stage leave [0,2p) and ONE conditional subtraction makes the residue canonical; the mod-down epilogue takes the lazy [0,3p) as it is
            inv_stages_lean<1, 5, true, true, 8, 64, !MD>(x, twA, pd.inv_n, (u32)pd.cr1, pc);
            if (!MD) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    u64 v[4] = {x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
                    csub4(v, pc.p);
#pragma unroll
                    for (int i = 0; i < 4; i++) x[4 * q + i] = v[i];
                }
            }
        } else {
        inv_stages<1, 5, true, true, 0>(x, twA, pd.inv_n, pc);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            u64 v[4] = {x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
            reduce4_from_4p(v, pc);
#pragma unroll
            for (int i = 0; i < 4; i++) x[4 * q + i] = v[i];
        }
        }
        if (MD) { // x = acc qk^-1 mod p (lean primes: lazily below 3p, else canonical): add the special limb's share and accumulate into the ciphertext (inner == 1 here)
            const unsigned t = opaque(tid);
            const Mod m = mod_of(pd);
            const u64 bias = pd.p * 4 + barrett64(a.md_half, m);         // [half]_p + 4p: keeps the difference below positive
            const u64 *special = a.data + (((u64)mm * period + a.md_dl) << N1_LOGN);
            u64 *dst = a.md_ct + (u64)(mm >> 1) * a.md_ct_bstride + (((u64)(mm & 1) * a.md_dl + slot) << N1_LOGN);
            const u64 *onto = !a.md_base ? dst : ((int)(mm & 1) >= a.md_base_polys) ? nullptr : a.md_base + (u64)(mm >> 1) * a.md_base_bstride + (((u64)(mm & 1) * a.md_dl + slot) << N1_LOGN);
            const Shoup iq[4] = {pd.aux, pd.aux, pd.aux, pd.aux};
#if N1_INV_TOP == 3
            TROY_WAIT_VMEM(); // the next row's staged half has landed (issued a round ago): waited for HERE, before this row's first store
#endif
#pragma unroll
            for (int g = 0; g < 8; g++) { // four coefficients at a time through the butterfly building blocks (bfly.h); x[] stays in registers
                u64 tl[4], c[4], q[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const unsigned n = t + 1024 * (4 * g + i);
                    tl[i] = ld_g(special, n) + a.md_half;                 // t' = (acc_last + half) mod qk
                    c[i] = onto ? ld_g(onto, n) : 0;
                }
                csub4(tl, a.md_qk);
                lite_reduce4(tl, (u32)pd.cr1, pc);                        // [t']_p lazily, below 4p
#pragma unroll
                for (int i = 0; i < 4; i++) tl[i] = bias - tl[i];         // [half]_p - [t']_p + 4p, in (0, 5p)
                mulhi_approx4_u(q, tl, iq);
#pragma unroll
                for (int i = 0; i < 4; i++) c[i] += mul_acc_u(x[4 * g + i], tl[i], pd.aux.op, q[i], pc.negp); // + acc qk^-1 (< 3p) + (..) qk^-1 (< 3p): below 7p
                csub4(c, pc.four_p);
                csub4(c, pc.two_p);
                csub4(c, pc.p);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    // (keeping the 32 results in x[] and storing them after the last operand load -- no load ever waited for behind a store -- spills 200-300 B:
                    // the epilogue's branches on `onto` leave the allocator no room.  The groups' own stores stay between their loads.)
                    st_g_inv<!FP && (N1_NT_INV & 2)>(dst, t + 1024 * (4 * g + i), c[i]);
                }
            }
        } else {
            const unsigned t = opaque(tid);
#if N1_INV_TOP == 3
            TROY_WAIT_VMEM(); // the next row's staged half has landed (issued a round ago): waited for HERE, while no store is in flight
#endif
#pragma unroll
            for (int r = 0; r < 32; r++)
                if (!(N1_INV_EXP & 2) || x[r] == 0x123456789abcdefull) st_g_inv<!FP && (N1_NT_INV & 1)>(row, t + 1024 * r, x[r]);
        }
    }
}
template <bool LEAN, bool MD> __global__ __launch_bounds__(N1_THREADS) void ntt1_inv_kernel(Ntt1Args a) { ntt1_inv_body<LEAN, MD, false>(a); }
template <bool MD> __global__ __launch_bounds__(N1_THREADS) void ntt1_inv_fp_kernel(Ntt1Args a) { ntt1_inv_body<true, MD, true>(a); }


// ---- N = 2^12 .. 2^14: the whole limb fits LDS (32 / 64 / 128 KiB), so the schedule is the same one without halves and parking.
// T = N / 16 threads (4 / 8 / 16 waves), 16 coefficients per thread: value (g, h) of thread t is coefficient t + T (h G + g), G = 16 / NSUB.
//   forward: round A = the first LOGA = LOGN - 10 stages on G groups of NSUB = 2^LOGA registers (workgroup-uniform twiddles), which splits the
//   limb into NSUB sub-blocks of 1024 points -- value (g, h) lands in sub-block h at position t + T g; one barrier; wave w finishes sub-block w
//   (rounds B, C1, C2 above: the same ten stages at every size) while the next row's sixteen loads are in flight;
//   inverse: the mirror image -- sub-blocks first (their inputs straight from HBM), one barrier, round A' with N^-1 folded in.
// Two to five workgroups share a compute unit (LDS), so the launch needs fewer rows than the N = 2^15 form to fill the chip.
template <int LOGN, bool LEAN, bool FP> __device__ __forceinline__ void ntt1s_fwd_body(const Ntt1Args &a) {
    constexpr int LOGA = LOGN - 10, NSUB = 1 << LOGA, T = 64 * NSUB, G = 16 >> LOGA;
    __shared__ __attribute__((aligned(16))) u64 lds[NSUB * 1024];
    const unsigned tid = threadIdx.x, lane = tid & 63;
#ifdef TROYHIP_CPU_EMUL
    const unsigned wv = tid >> 6;
#else
    const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    unsigned sidx, chunk;
    if (!n1_unit(a, sidx, chunk)) return;
    const unsigned slot = uniform_u32(a.slots[sidx]);
    PrimeDesc pd = a.primes[uniform_u32(a.map.id[slot])];
    if constexpr (FP) pd.root = pd.root_fp;
    const FpPrime fc = make_fp_prime_uniform(FP ? pd.p : 1);
    const PrimeConst pc = make_prime_const(pd.p);
    const Mod m = mod_of(pd);
    const unsigned m_begin = chunk * a.rows_per_wg;
    const unsigned m_end = (m_begin + a.rows_per_wg < a.m_total) ? m_begin + a.rows_per_wg : a.m_total;
    const unsigned inner = a.map.inner, period = a.map.period;
    auto row_of = [&](unsigned mm) -> u64 {
        const unsigned o = mm / inner, k = mm - o * inner;
        return (((u64)o * period + slot) * inner + k) << LOGN;
    };
    const u64 *in_base = a.src ? a.src : a.data;
    u64 *const region = lds + 1024 * wv;
    u64 y[16]; // y[(g << LOGA) + h]
    auto load_row = [&](const u64 *rowp) {
        const unsigned t = opaque(tid);
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int h = 0; h < NSUB; h++) y[(g << LOGA) + h] = ld_g_fwd(rowp, t + T * (h * G + g));
    };
    load_row(in_base + row_of(m_begin));
    const Ntt1Args &args = a;
    for (unsigned mm = m_begin; mm < m_end; mm++) {
        u64 *const out = a.data + row_of(mm);
        if constexpr (FP) {
#pragma unroll
            for (int r = 0; r < 16; r++) y[r] = fp_bits(fp_from_u64(y[r]));
            if (a.fp_red_mask & 1u) fp_reduce_all<16>(y, fc);
        }
        auto twA = [&](int st, int, int blk) { return ld_tw_uniform((pd.root + (1u << st) + blk)); };
        if constexpr (FP) fp_fwd_stages<G, LOGA>(y, twA, fc);
        else fwd_stages<G, LOGA, LEAN, true>(y, twA, pc);
        if (mm != m_begin) __syncthreads(); // every wave is done with its region's previous content
        {
            const unsigned t = opaque(tid);
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int h = 0; h < NSUB; h++) lds[1024 * h + sw1(t + T * g)] = y[(g << LOGA) + h];
        }
        __syncthreads();
        if (mm + 1 < m_end) load_row(in_base + row_of(mm + 1)); // all sixteen registers are free: the next limb streams in under the sub-block rounds
        fwd_subblock<LOGN, LEAN, false, FP>(lds, 1024 * wv, wv, lane, pd, pc, m, out + 1024 * wv, args, mm, m_begin, 4, true, fc);
    }
}
template <int LOGN, bool LEAN> __global__ __launch_bounds__(64 << (LOGN - 10), 4) void ntt1s_fwd_kernel(Ntt1Args a) { ntt1s_fwd_body<LOGN, LEAN, false>(a); }
template <int LOGN> __global__ __launch_bounds__(64 << (LOGN - 10), 4) void ntt1s_fwd_fp_kernel(Ntt1Args a) { ntt1s_fwd_body<LOGN, true, true>(a); }

template <int LOGN, bool LEAN, bool MD, bool FP> __device__ __forceinline__ void ntt1s_inv_body(const Ntt1Args &a) {
    constexpr int LOGA = LOGN - 10, NSUB = 1 << LOGA, T = 64 * NSUB, G = 16 >> LOGA;
    constexpr unsigned NN = 1u << LOGN;
    __shared__ __attribute__((aligned(16))) u64 lds[NSUB * 1024];
    const unsigned tid = threadIdx.x, lane = tid & 63;
#ifdef TROYHIP_CPU_EMUL
    const unsigned wv = tid >> 6;
#else
    const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    unsigned sidx, chunk;
    if (!n1_unit(a, sidx, chunk)) return;
    const unsigned slot = uniform_u32(a.slots[sidx]);
    PrimeDesc pd = a.primes[uniform_u32(a.map.id[slot])];
    if constexpr (FP) { pd.iroot = pd.iroot_fp; pd.inv_n = pd.inv_n_fp; pd.iroot_last_scaled = pd.iroot_last_scaled_fp; }
    const FpPrime fc = make_fp_prime_uniform(FP ? pd.p : 1);
    const PrimeConst pc = make_prime_const(pd.p);
    const unsigned m_begin = chunk * a.rows_per_wg;
    const unsigned m_end = (m_begin + a.rows_per_wg < a.m_total) ? m_begin + a.rows_per_wg : a.m_total;
    const unsigned inner = a.map.inner, period = a.map.period;
    auto row_of = [&](unsigned mm) -> u64 {
        const unsigned o = mm / inner, k = mm - o * inner;
        return (((u64)o * period + slot) * inner + k) << LOGN;
    };
    u64 *const region = lds + 1024 * wv;
    u64 y[16]; // the wave's sub-block input: y[8 i + r] = coefficient 8 (lane + 64 i) + r
    auto load16 = [&](const u64 *sub) {
        const unsigned l = opaque(lane);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const ulonglong2 v = ld_g2(sub, 8 * (l + 64 * i) + 2 * q);
                y[8 * i + 2 * q] = v.x;
                y[8 * i + 2 * q + 1] = v.y;
            }
    };
    load16(a.data + row_of(m_begin) + 1024 * wv);
    for (unsigned mm = m_begin; mm < m_end; mm++) {
        u64 *const row = a.data + row_of(mm);
        if (mm != m_begin) __syncthreads(); // round A' of the previous row has read the regions
        inv_subblock<LOGN, LEAN, FP>(y, lds, 1024 * wv, wv, lane, pd, pc, fc, a.fp_red_mask);
        __syncthreads();
        u64 x[16]; // x[(g << LOGA) + h] = coefficient t + T g of sub-block h after its ten stages
        {
            const unsigned t = opaque(tid);
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int h = 0; h < NSUB; h++) x[(g << LOGA) + h] = lds[1024 * h + sw2(t + T * g)];
        }
        if (mm + 1 < m_end) load16(a.data + row_of(mm + 1) + 1024 * wv); // the next row's sub-block input, in flight under round A'
        auto twA = [&](int st, int, int blk) { return st == LOGA - 1 ? pd.iroot_last_scaled : ld_tw_uniform((pd.iroot + (NN - ((unsigned)NSUB >> st) + 1) + blk)); };
        if constexpr (FP) {
            if (a.fp_red_mask & 8u) fp_reduce_all<16>(x, fc);
            if constexpr (LOGA > 1) fp_inv_stages<G, LOGA, true, 0, LOGA - 1>(x, twA, pd.inv_n, fc);
            if (a.fp_red_mask & 16u) fp_reduce_all<16>(x, fc);
            fp_inv_stages<G, LOGA, true, LOGA - 1, LOGA>(x, twA, pd.inv_n, fc);
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = fp_canonical(fp_of_bits(x[i]), fc, pd.p);
        } else if constexpr (LEAN) { // as in the N = 2^15 form: 8p in, per-register bounds, exact last quotients for plain stores
            inv_stages_lean<G, LOGA, true, true, 8, 64, !MD>(x, twA, pd.inv_n, (u32)pd.cr1, pc);
            if (!MD) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    u64 v[4] = {x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
                    csub4(v, pc.p);
#pragma unroll
                    for (int i = 0; i < 4; i++) x[4 * q + i] = v[i];
                }
            }
        } else {
            inv_stages<G, LOGA, true, true, 0>(x, twA, pd.inv_n, pc);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                u64 v[4] = {x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
                reduce4_from_4p(v, pc);
#pragma unroll
                for (int i = 0; i < 4; i++) x[4 * q + i] = v[i];
            }
        }
        const unsigned t = opaque(tid);
        if (MD) { // the BFV mod-down by the special prime as the store epilogue (Ntt1ModDown; the N = 2^15 form above has the derivation)
            const Mod m = mod_of(pd);
            const u64 bias = pd.p * 4 + barrett64(a.md_half, m);
            const u64 *special = a.data + (((u64)mm * period + a.md_dl) << LOGN);
            u64 *dst = a.md_ct + (u64)(mm >> 1) * a.md_ct_bstride + (((u64)(mm & 1) * a.md_dl + slot) << LOGN);
            const u64 *onto = !a.md_base ? dst : ((int)(mm & 1) >= a.md_base_polys) ? nullptr : a.md_base + (u64)(mm >> 1) * a.md_base_bstride + (((u64)(mm & 1) * a.md_dl + slot) << LOGN);
            const Shoup iq[4] = {pd.aux, pd.aux, pd.aux, pd.aux};
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) {
                u64 tl[4], c[4], q[4];
                unsigned n[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int idx = 4 * q4 + i, g = idx >> LOGA, h = idx & (NSUB - 1);
                    n[i] = t + T * (h * G + g);
                    tl[i] = ld_g(special, n[i]) + a.md_half;
                    c[i] = onto ? ld_g(onto, n[i]) : 0;
                }
                csub4(tl, a.md_qk);
                lite_reduce4(tl, (u32)pd.cr1, pc);
#pragma unroll
                for (int i = 0; i < 4; i++) tl[i] = bias - tl[i];
                mulhi_approx4_u(q, tl, iq);
#pragma unroll
                for (int i = 0; i < 4; i++) c[i] += mul_acc_u(x[4 * q4 + i], tl[i], pd.aux.op, q[i], pc.negp);
                csub4(c, pc.four_p);
                csub4(c, pc.two_p);
                csub4(c, pc.p);
#pragma unroll
                for (int i = 0; i < 4; i++) st_g(dst, n[i], c[i]);
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int h = 0; h < NSUB; h++) st_g(row, t + T * (h * G + g), x[(g << LOGA) + h]);
        }
    }
}
template <int LOGN, bool LEAN, bool MD> __global__ __launch_bounds__(64 << (LOGN - 10), 4) void ntt1s_inv_kernel(Ntt1Args a) { ntt1s_inv_body<LOGN, LEAN, MD, false>(a); }
template <int LOGN, bool MD> __global__ __launch_bounds__(64 << (LOGN - 10), 4) void ntt1s_inv_fp_kernel(Ntt1Args a) { ntt1s_inv_body<LOGN, true, MD, true>(a); }

// ---- host side ----

unsigned device_cus() { // of the calling thread's current device = the context's (every entry point binds it: capi.cpp need()); one query per device
    static std::atomic<unsigned> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    std::atomic<unsigned> &slot = cus[(unsigned)dev & 63u];
    unsigned n = slot.load(std::memory_order_relaxed);
    if (!n) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 256; }
        n = (unsigned)v;
        slot.store(n, std::memory_order_relaxed);
    }
    return n;
}
// TROYHIP_NTT = twopass | single forces one form (tests, A/B runs); by default the single-pass kernel takes the launches that give
// every CU at least four rows.  Below that a whole-limb workgroup per CU is the wrong grain -- a lone row costs ~45 us however few rows
// there are, while the two-pass kernels spread a row over 32 small workgroups: BFV multiply+relinearize at B = 1 takes 0.29 ms with
// two passes and 0.45 ms with one, the two meet at B = 32 (1000-1400 rows per launch), and from there the single pass wins.
static int ntt1_mode() {
    static const int mode = [] {
        const char *e = std::getenv("TROYHIP_NTT");
        return !e ? 0 : (std::strcmp(e, "twopass") == 0 ? 1 : (std::strcmp(e, "single") == 0 ? 2 : 0));
    }();
    return mode;
}
// workgroups of the single-pass kernels that share a compute unit: one at N = 2^15 / 2^14 (LDS), two at 2^13, four at 2^12 (16 waves)
static unsigned ntt1_wgs_per_cu(int logn) { return logn >= 14 ? 1u : (logn == 13 ? 2u : 4u); }
// feature: 0 the plain transform (in place, or forward from a source of the same layout), 1 the BFV mod-down epilogue, 2 the CKKS correction form,
// 3 the inverse from a strided source.  N = 2^15 has them all, the smaller sizes (ntt1s_*) the first two.
bool ntt1_supported(int logn, const LimbMap &map, size_t rows, int feature) {
    if (ntt1_mode() == 1 || logn < 12 || logn > N1_LOGN || !rows || rows % ((size_t)map.period * map.inner)) return false;
    if (logn != N1_LOGN && feature > 1) return false;
    // the launch must give every workgroup slot of the chip about four rows: below that the per-row latency of a whole-limb workgroup shows and the
    // two-pass kernels, which spread a row over N / 2048 small workgroups, are the better grain
    return ntt1_mode() == 2 || rows >= 4 * (size_t)device_cus() * ntt1_wgs_per_cu(logn);
}
// rows laid out r = (o * period + i) * inner + k, prime map.id[i].  The forward transform is launched once per prime class:
// guard-free butterflies for the slots in map.lean, guarded ones for the rest.
void launch_ntt1(u64 *data, const u64 *src, const PrimeDesc *primes, const LimbMap &map, size_t rows, int logn, bool inverse, hipStream_t stream, u64 slot_mask,
                 const Ntt1ModDown *md, const Ntt1Corr *cr, u64 src_ostride) {
    if (rows == 0) return;
    if (logn < 12 || logn > N1_LOGN) throw Error(ST_LOGIC_ERROR, "ntt1: unsupported size");
    if (logn != N1_LOGN && (cr || (src && inverse))) throw Error(ST_LOGIC_ERROR, "ntt1: the correction form and the strided source belong to N = 2^15");
    const size_t per_outer = (size_t)map.period * map.inner;
    if (rows % per_outer) throw Error(ST_INVALID_ARGUMENT, "ntt1: row count must be a multiple of the limb pattern");
    if (md && (!inverse || map.inner != 1 || md->dl >= map.period)) throw Error(ST_LOGIC_ERROR, "ntt1: the mod-down epilogue belongs to the inverse transform of [..][slot][N] accumulators");
    Ntt1Args a;
    std::memset(&a, 0, sizeof(a));
    a.data = data;
    a.src = src;
    a.src_ostride = src_ostride;
    if (src && inverse && !src_ostride) throw Error(ST_LOGIC_ERROR, "ntt1: the out-of-place inverse takes a strided source");
    a.primes = primes;
    a.map = map;
    a.m_total = (unsigned)(rows / per_outer * map.inner);
    if (cr) {
        if (inverse || map.inner != 1 || src) throw Error(ST_LOGIC_ERROR, "ntt1: the correction form belongs to the forward transform of [outer][slot][N] rows");
        a.cr_last = cr->last; a.cr_in = cr->in; a.cr_out = cr->out; a.cr_inv = cr->inv;
        a.cr_in_ostride = cr->in_ostride; a.cr_in_gstride = cr->in_gstride; a.cr_out_gstride = cr->out_gstride; a.cr_out_ostride = cr->out_ostride;
        a.cr_qx = cr->qx; a.cr_half = cr->half; a.cr_group = cr->group ? cr->group : 1; a.cr_accumulate = cr->accumulate ? 1 : 0;
        a.cr_base = cr->base; a.cr_base_gstride = cr->base_gstride; a.cr_base_polys = cr->base_polys;
    }
    if (md) { a.md_ct = md->ct; a.md_ct_bstride = md->ct_bstride; a.md_qk = md->qk; a.md_half = md->half; a.md_dl = (unsigned)md->dl; a.md_base = md->base; a.md_base_bstride = md->base_bstride; a.md_base_polys = md->base_polys; }
    // One workgroup fills a CU, so a launch runs in rounds of `cus` workgroups and a round lasts as long as a workgroup's rows (plus
    // the un-overlapped first load and last store, about a third of a row): pick the rows per workgroup (they share the prime) that
    // minimises rounds x (rows + 1/3), with a small penalty for spreading the CUs over many primes at once.  A fixed "three workgroups
    // per CU" left mid-size launches with a mostly idle last round.
    const unsigned cus = device_cus() * ntt1_wgs_per_cu(logn); // workgroup slots of the chip
    static const unsigned forced_rpw = [] { const char *e = probe_env("TROYHIP_NTT1_RPW"); return e ? (unsigned)std::atoi(e) : 0u; }(); // tests: row loop at small batches
    static const int forced_xcd = [] { const char *e = probe_env("TROYHIP_NTT1_XCD"); return e ? std::atoi(e) : -1; }();
    static const int forced_group = [] { const char *e = probe_env("TROYHIP_NTT1_XCD_GROUP"); return e ? std::atoi(e) : 0; }();
    auto plan = [&](unsigned nslots) { // -> rows per workgroup for a launch over `nslots` primes
        if (forced_rpw) return forced_rpw;
        unsigned best = 1;
        double best_cost = 1e30;
        for (unsigned r = 1; r <= 32 && r <= a.m_total; r++) {
            const unsigned per_prime = (a.m_total + r - 1) / r, wgs = nslots * per_prime;
            // workgroups of one prime are adjacent in the grid: fewer of them per prime means more primes in flight at once, whose
            // twiddle tables (1 MiB each) then compete for the 4 MiB L2 of an XCD (PMC: +17 % fetch traffic at 4.4 primes in flight)
            const double in_flight = per_prime < cus ? (double)cus / per_prime : 1.0;
            const double cost = (double)((wgs + cus - 1) / cus) * (r + 0.34) * (1.0 + 0.008 * (in_flight - 1.0));
            if (cost < best_cost - 1e-9) { best_cost = cost; best = r; }
        }
        return best;
    };
    // split by prime class (guard-free / guarded butterflies), back to back on the caller's stream; the lean inverse takes canonical
    // inputs (every stored limb is).  Tried and rejected: both classes in one kernel behind a workgroup-uniform branch (the merged
    // function spills 40-120 registers), the smaller class forked onto a companion stream (concurrent kernels interleave their
    // workgroups and lose the per-prime locality: 0.34 -> 0.28 at B=64).
    // a third class since round 3: the FP64 instances for the primes in [2^33, 2^50) (map.fp within map.lean; fpmod.h), when the map
    // carries the host prime list the bound walk needs
    struct Cls { Ntt1Args a; bool lean, fp; } cls[3];
    int ncls = 0;
    const u64 fp_slots = map.host_primes ? (map.fp & map.lean) : 0;
    for (int kind = 2; kind >= 0; kind--) { // 2: FP64, 1: guard-free integer, 0: guarded integer
        a.nslots = 0;
        u64 pmax = 0;
        for (unsigned i = 0; i < map.period; i++) {
            if (!((slot_mask >> i) & 1)) continue;
            const int k = ((fp_slots >> i) & 1) ? 2 : (int)((map.lean >> i) & 1);
            if (k != kind) continue;
            a.slots[a.nslots++] = (uint8_t)i;
            if (kind == 2) pmax = std::max(pmax, map.host_primes[map.id[i]]);
        }
        if (!a.nslots) continue;
        a.fp_red_mask = 0;
        if (kind == 2) { // where the values are reduced (fpmod.h): forward rounds A, B, C1, C2 from canonical inputs (the correction form starts below 5p);
                         // inverse rounds D', C', B', the first four stages of A', its last stage
            const int la = logn - 10; // stages of the cross-sub-block round
            const int fwd_rounds[4] = {la, 4, 3, 3}, inv_rounds[5] = {3, 3, 4, la - 1, 1};
            const FpPlan pl = inverse ? fp_plan_inv(pmax, 1.0, inv_rounds, 5) : fp_plan(pmax, cr ? 5.0 : 1.0, fwd_rounds, 4, 1.5);
            if (pl.out_bound < 0) throw Error(ST_LOGIC_ERROR, "ntt1: FP64 bound walk");
            a.fp_red_mask = pl.mask;
        }
        stats::counter(kind == 2 ? stats::NTT1_FP_LAUNCHES : stats::NTT1_INT_LAUNCHES).fetch_add(1, std::memory_order_relaxed);
        a.rows_per_wg = plan(a.nslots);
        a.chunks = (a.m_total + a.rows_per_wg - 1) / a.rows_per_wg;
        // one prime, or a grid the chip holds at once: nothing to separate (probe builds: TROYHIP_NTT1_XCD = 1 always / 0 never, for the tests)
        // The forms whose rows of different primes share a row (mod-down: the special limb; divide-and-round: the dropped limb) take the list -- grouped, below -- as
        // soon as the launch is more than ONE round of workgroups: at the headline's two lanes of 128 the mod-down is 481 workgroups on 256 CUs, and flat order
        // fetched the special limb once per prime (same-box A/B, profiles/r06_xcd_md_ab.txt: that kernel 867 -> 853 us, headline +0.4 %)
        const bool shared_row = a.md_ct || a.cr_last;
        const bool xcd = forced_xcd >= 0 ? forced_xcd != 0 : (N1_XCD && a.nslots > 1 && a.nslots * a.chunks >= (shared_row ? cus + 1 : 2 * cus));
        a.xcd_per = xcd ? (a.nslots * a.chunks + 7) / 8 : 0;
        a.xcd_group = forced_group > 0 ? (unsigned)forced_group : !(a.md_ct || a.cr_last) ? 1u : kind == 2 ? N1_XCD_GROUP_FP : N1_XCD_GROUP_INT;
        static const bool perturb = probe_env("TROYHIP_NTT1_XCD_PERTURB") != nullptr;
        a.xcd_perturb = perturb ? 1u : 0u;
        cls[ncls].a = a;
        cls[ncls].lean = kind != 0;
        cls[ncls++].fp = kind == 2;
    }
    auto launch_small = [&](const Cls &k, hipStream_t st, auto logn_tag) { // N = 2^12 .. 2^14 (ntt1s_*)
        constexpr int LN = decltype(logn_tag)::value;
        const Ntt1Args &x = k.a;
        const dim3 grid(x.xcd_per ? 8 * x.xcd_per : x.nslots * x.chunks), block(64u << (LN - 10));
#ifndef TROYHIP_CPU_EMUL
        if (ktime::enabled) { // the instance's name as rocprofv3 prints it (the launch macro would say "LN")
            static thread_local char tagbuf[64];
            if (k.fp) std::snprintf(tagbuf, sizeof(tagbuf), inverse ? "ntt1s_inv_fp_kernel<%d, %s>" : "ntt1s_fwd_fp_kernel<%d>", LN, x.md_ct ? "true" : "false");
            else if (inverse) std::snprintf(tagbuf, sizeof(tagbuf), "ntt1s_inv_kernel<%d, %s, %s>", LN, k.lean ? "true" : "false", x.md_ct ? "true" : "false");
            else std::snprintf(tagbuf, sizeof(tagbuf), "ntt1s_fwd_kernel<%d, %s>", LN, k.lean ? "true" : "false");
            ktime::tag = tagbuf;
        }
#endif
        if (k.fp) {
            if (!inverse) TROY_LAUNCH(HIP_KERNEL_NAME(ntt1s_fwd_fp_kernel<LN>), grid, block, 0, st, x);
            else if (x.md_ct) TROY_LAUNCH(HIP_KERNEL_NAME(ntt1s_inv_fp_kernel<LN, true>), grid, block, 0, st, x);
            else TROY_LAUNCH(HIP_KERNEL_NAME(ntt1s_inv_fp_kernel<LN, false>), grid, block, 0, st, x);
        } else if (!inverse) {
            if (k.lean) TROY_LAUNCH(HIP_KERNEL_NAME(ntt1s_fwd_kernel<LN, true>), grid, block, 0, st, x);
            else TROY_LAUNCH(HIP_KERNEL_NAME(ntt1s_fwd_kernel<LN, false>), grid, block, 0, st, x);
        } else if (x.md_ct) {
            if (k.lean) TROY_LAUNCH(HIP_KERNEL_NAME(ntt1s_inv_kernel<LN, true, true>), grid, block, 0, st, x);
            else TROY_LAUNCH(HIP_KERNEL_NAME(ntt1s_inv_kernel<LN, false, true>), grid, block, 0, st, x);
        } else if (k.lean) TROY_LAUNCH(HIP_KERNEL_NAME(ntt1s_inv_kernel<LN, true, false>), grid, block, 0, st, x);
        else TROY_LAUNCH(HIP_KERNEL_NAME(ntt1s_inv_kernel<LN, false, false>), grid, block, 0, st, x);
    };
    auto launch = [&](const Cls &k, hipStream_t st) {
        if (logn == 14) { launch_small(k, st, std::integral_constant<int, 14>{}); return; }
        if (logn == 13) { launch_small(k, st, std::integral_constant<int, 13>{}); return; }
        if (logn == 12) { launch_small(k, st, std::integral_constant<int, 12>{}); return; }
        const Ntt1Args &x = k.a;
        const dim3 grid(x.xcd_per ? 8 * x.xcd_per : x.nslots * x.chunks);
        if (k.fp) {
            if (inverse) {
                if (x.md_ct) TROY_LAUNCH(HIP_KERNEL_NAME(ntt1_inv_fp_kernel<true>), grid, dim3(N1_THREADS), 0, st, x);
                else TROY_LAUNCH(HIP_KERNEL_NAME(ntt1_inv_fp_kernel<false>), grid, dim3(N1_THREADS), 0, st, x);
            } else if (x.cr_last) TROY_LAUNCH(HIP_KERNEL_NAME(ntt1_fwd_fp_kernel<true>), grid, dim3(N1_THREADS), 0, st, x);
            else TROY_LAUNCH(HIP_KERNEL_NAME(ntt1_fwd_fp_kernel<false>), grid, dim3(N1_THREADS), 0, st, x);
            return;
        }
        if (inverse) {
            if (x.md_ct) {
                if (k.lean) TROY_LAUNCH(HIP_KERNEL_NAME(ntt1_inv_kernel<true, true>), grid, dim3(N1_THREADS), 0, st, x);
                else TROY_LAUNCH(HIP_KERNEL_NAME(ntt1_inv_kernel<false, true>), grid, dim3(N1_THREADS), 0, st, x);
            } else if (k.lean) TROY_LAUNCH(HIP_KERNEL_NAME(ntt1_inv_kernel<true, false>), grid, dim3(N1_THREADS), 0, st, x);
            else TROY_LAUNCH(HIP_KERNEL_NAME(ntt1_inv_kernel<false, false>), grid, dim3(N1_THREADS), 0, st, x);
        } else {
            if (x.cr_last) {
                if (k.lean) TROY_LAUNCH(HIP_KERNEL_NAME(ntt1_fwd_kernel<true, true>), grid, dim3(N1_THREADS), 0, st, x);
                else TROY_LAUNCH(HIP_KERNEL_NAME(ntt1_fwd_kernel<false, true>), grid, dim3(N1_THREADS), 0, st, x);
            } else if (k.lean) TROY_LAUNCH(HIP_KERNEL_NAME(ntt1_fwd_kernel<true, false>), grid, dim3(N1_THREADS), 0, st, x);
            else TROY_LAUNCH(HIP_KERNEL_NAME(ntt1_fwd_kernel<false, false>), grid, dim3(N1_THREADS), 0, st, x);
        }
    };
#ifdef N1_TIMING // development build: phase stamps of one workgroup of the guard-free forward kernel
    static u64 *dbg = nullptr;
    if (!dbg) HIP_CHECK(hipMalloc((void **)&dbg, 16 * 8 * 8));
    for (int i = 0; i < ncls; i++)
        if (cls[i].lean && !inverse) { cls[i].a.dbg = dbg; cls[i].a.dbg_block = (cls[i].a.nslots * cls[i].a.chunks) / 2 + 3; }
#endif
    for (int i = 0; i < ncls; i++) launch(cls[i], stream);
    launch_check("ntt1 kernels");
#ifdef N1_TIMING
    if (!inverse && ncls && cls[0].lean) {
        u64 h[16 * 8];
        HIP_CHECK(hipStreamSynchronize(stream));
        HIP_CHECK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        for (unsigned r = 0; r < cls[0].a.rows_per_wg && r < 8; r++) {
            fprintf(stderr, "n1 fwd row %u:", r);
            for (int i = 1; i < 14; i++) fprintf(stderr, " %lld", (long long)(h[16 * r + i] - h[16 * r + i - 1]));
            if (r + 1 < cls[0].a.rows_per_wg && r + 1 < 8) fprintf(stderr, " | next %lld", (long long)(h[16 * (r + 1)] - h[16 * r + 13]));
            fprintf(stderr, "\n");
        }
    }
#endif
}

} // namespace troyhip
