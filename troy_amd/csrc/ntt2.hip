// ntt2.hip -- the production NTT/INTT for N >= 4096 on gfx950 ("twiddle-stationary" two-pass transform).
//
// Same transform and tables as ntt.hip (which stays as the generic path for N < 4096); what changes is how a
// pass is executed on a CU:
//   * the stage plan is a template parameter (NS stages, LOGC column bits): every index, gap and LDS address of a
//     round is a compile-time expression;
//   * a 256-thread workgroup owns one 2048-coefficient tile POSITION and loops over up to `rows_per_wg` limb-rows
//     that share the same prime (the polynomials of a batch, or the L digits of one key-switch output prime), so
//     the twiddles of all rounds are loaded ONCE into registers/SGPRs and reused -- twiddle traffic (2x the data in
//     the last stages, SURVEY section 7 "hard parts") drops by rows_per_wg;
//   * the first round reads its 8 points per thread straight from HBM and the last round writes straight back
//     (no staging pass through LDS); between rounds the tile is exchanged through LDS with an XOR swizzle
//     addr = f ^ ((f>>3)&7) ^ (((f>>6)&3)<<3) that is bank-conflict-free for every butterfly distance;
//   * strided passes double-buffer the LDS exchange across rows (2 barriers per row instead of one per stage); contiguous
//     passes need no workgroup barrier at all (every 512-point sub-transform stays in one wave) and stream the next row
//     into a staging area with LDS-DMA (global_load_lds_dwordx4) while the current one is transformed;
//   * the strided pass can read its input from another buffer: either the same rows of another allocation (operands are
//     consumed in place) or the digit decomposition of key switching (evaluator.cpp:2432-2442, the reference's
//     kModuloPolyCoeffs + copy per (i,j)), Barrett-reducing limb k modulo the row's prime on the fly;
//   * the last forward pass has two fused epilogues (2 waves per SIMD): MAC = 1 accumulates the transforms of the L digits
//     against the key-switching key (the expanded digits are never stored transformed), MAC = 2 keeps (a0, a1, b0, b1) and
//     stores the BEHZ ciphertext tensor.
#include "kernels.h"
#include "bfly.h"
#include "fpmod.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace troyhip {

// tools/ntt_probe.sh builds throw-away variants with parts of the kernel removed to see what bounds it (results are
// wrong by construction): bit0 no HBM traffic, bit1 no LDS exchange, bit2 no butterflies, bit3 the contiguous pass reads an
// L2-resident window instead of its rows.  Always 0 in the product.
#ifndef N2_EXP
#define N2_EXP 0
#endif
#ifndef N2_MIN_WAVES
#define N2_MIN_WAVES 4
#endif
#ifndef N2_MAC_WAVES
#define N2_MAC_WAVES 2 // waves per SIMD of the key-switch fused kernel (16 x 128-bit accumulators per thread)
#endif
#ifndef N2_MAC_HOIST
#define N2_MAC_HOIST 1 // keep the last round's twiddles in registers in that kernel
#endif
#ifndef N2_DMA
#define N2_DMA 1 // contiguous passes prefetch the next row with LDS-DMA
#endif
#ifndef N2_DMA_POL
#define N2_DMA_POL 2 // cache policy of the LDS-DMA row prefetch: 2 = non-temporal -- rows that are read once do not displace the key window (and the staged rows of
                     // the neighbours) in L2: fetch of the key-switch accumulating pass 15.98 -> 14.25 GB per 256 ciphertexts (1.10x -> 1.00x its rows), of the tensor
                     // pass 8.06 -> 7.89; -0.8 % / -2 % on the two kernels, +0.3-0.5 % on the 49-bit twin, the CKKS chain and BGV (profiles/r05_inv_probes.txt 5e).  0: default policy
#endif
#ifndef N2_MAC_STORE_LINEAR
#define N2_MAC_STORE_LINEAR 1 // the key-switch sums are stored through the wave's exchange area (contiguous KiB per instruction); 0 (probe): from a thread's eight consecutive coefficients
#endif
#ifndef N2_COALESCED_STORE
#define N2_COALESCED_STORE 1 // forward contiguous pass: transpose the last round through LDS, store 1 KiB per instruction
#endif
#ifndef N2_TENSOR_WAVES
#define N2_TENSOR_WAVES 3 // waves per SIMD of the tensor fused kernel: with the LDS addresses formed per row (N2_FRESH_TENSOR) it fits 168 VGPRs
#endif                    // (196 with hoisted addresses = 2 waves): 1183 -> 1044 us per step, headline +3 %
#ifndef N2_TENSOR_HOIST1
#define N2_TENSOR_HOIST1 1 // tensor fused kernel: keep the per-lane twiddles of round 1 / round 2 in registers (0: re-read them per row)
#endif
#ifndef N2_TENSOR_HOIST2
#define N2_TENSOR_HOIST2 1
#endif
#ifndef N2_FRESH_TENSOR
#define N2_FRESH_TENSOR 15
#endif
#ifndef N2_FRESH_MD
#define N2_FRESH_MD 15
#endif
#ifndef N2_FRESH_FP_MAC
#define N2_FRESH_FP_MAC 15
#endif
#ifndef N2_FP_MAC_WAVES
#define N2_FP_MAC_WAVES 3 // waves per SIMD of the FP64 key-switch fused kernel (16 doubles of accumulators per thread)
#endif
#define N2_THREADS 256
#define N2_LOGT 11
#define N2_T 2048

struct Ntt2Args {
    u64 *data;            // destination (and source when src == nullptr): rows of N coefficients
    const u64 *src;       // optional distinct source for the strided forward pass (key-switch fusion)
    u64 src_ostride;      // words between consecutive `outer` items of src
    const PrimeDesc *primes;
    LimbMap map;          // row r = (o * period + i) * inner + k  has prime map.id[i]
    int logn;
    unsigned tiles_per_row_log;
    unsigned m_total;     // outer * inner rows per prime slot
    unsigned rows_per_wg; // R
    unsigned chunks;      // ceil(m_total / R)
    int src_reduce;       // reduce src values modulo the row prime (they are residues of another prime)
    u64 src_bound;        // exclusive upper bound of the src values (largest source prime): rows whose prime p has 8p > bound skip the reduction
    int slot_fastest;     // workgroup order, see the kernel
    int src_same_layout;  // 1: src has the row layout of data (plain out-of-place transform) instead of the digit broadcast
                          // 2: only the first src_slots prime slots of an item are read from an operand ([item][src_slots][N]: items below src_split from src,
                          //    the others from src2); the remaining slots are transformed in place (BEHZ multiply, both bases in one launch)
    const u64 *src2;
    unsigned src_slots, src_split;
    // key-switch inner product fused into the last forward pass (MAC = 1): the workgroup's rows are the dl digits of one
    // (ciphertext o, output prime slot); instead of storing the transforms it accumulates  sum_k NTT(d_k) (.) key[k][c][slot]
    // MAC = 2: ciphertext tensor fused into the last forward pass: the workgroup's four rows are (a0, a1, b0, b1) of one
    // (ciphertext, limb); their transforms stay in registers and only d0 = a0 b0, d1 = a0 b1 + a1 b0, d2 = a1 b1 are stored
    u64 *tensor_b;          // second operand, same layout as data ([batch][2][period][N])
    u64 *tensor_out;        // [batch][3][period][N]
    const u64 *mac_key;     // [dl][2][K][N]
    u64 *mac_acc;           // [outer][2][period][N]
    const u64 *mac_target;  // CKKS: the NTT-form input supplies the (k == slot) operand (evaluator.cpp:2424-2427), else nullptr
    u64 mac_tstride;
    unsigned mac_K;
    int mac_lazy;           // every output prime p satisfies dl * 8p * p < 2^128: the transforms are accumulated unreduced
    uint8_t mac_key_limb[65];
    // inverse transforms of a range of prime slots only, and the key-switch mod-down as the store epilogue of the last pass
    // (Ntt2ModDown, kernels.h; FINAL = 3 BFV / 4 BGV): rows are acc[o][slot][N] with inner == 1, o = 2 b + k
    unsigned slot_begin = 0;
    u64 *md_ct = nullptr;
    u64 md_ct_bstride = 0, md_qk = 0, md_half = 0;
    const u64 *md_share = nullptr; // BGV: [o][N] 128-bit integers al + k_t qk (ks_bgv_share_kernel, poly.hip)
    const u64 *md_base = nullptr;  // not null: accumulate onto (base[b], 0) instead of onto what ct holds
    u64 md_base_bstride = 0;
    int md_base_polys = 1;         // 2: component 1 reads base[b] + dl * N too (relinearize out of place)
    int skip_diag = 0;             // CKKS key switch: the (digit k == output slot) rows are not expanded -- the second pass takes them from mac_target
    unsigned md_dl = 0;
    // the prime slots this launch covers: workgroup group index -> slot = sel[index] (a launch per prime class: the FP64 instances take the
    // primes below 2^50, the integer instances the rest; partial inverse launches take a range)
    uint8_t sel[64];
    unsigned nsel = 0;
    // FP64 instances (ntt2_fp_kernel, fpmod.h)
    u64 fp_src_wide = 0;       // bit k: the residues of source digit k do not fit a double exactly (prime >= 2^50): integer Barrett step before the conversion
    unsigned fp_red_mask = 0;  // bit r: the values are reduced before round r of this pass (fp_plan)
    unsigned fp_acc_every = 0; // the key-switch accumulators are reduced every this many rows (0: never)
};

#ifndef N2_NT
#define N2_NT 1 // non-temporal accesses.  bit 0 (ON): the stores of the forward STRIDED pass -- rows written once and read back by another kernel: the key switch's
                // digit-expanding pass -2.5 %, headline +0.6 %, 49-bit twin +2 %, CKKS chain +1.3 %, BGV +0.6 % (profiles/r05_inv_probes.txt 5f).  Probes: bit 1 its loads
                // (the L + 1 readers of a source digit lose their L2 hits: slower), bit 2 the inverse strided pass's stores, bit 3 the contiguous pass's stores
                // (store_via_lds: transforms, tensor), bit 4 the key-switch sums
#endif
template <bool NT> __device__ __forceinline__ u64 n2_ld(const u64 *p) {
#ifndef TROYHIP_CPU_EMUL
    if (NT) return __builtin_nontemporal_load(p);
#endif
    return *p;
}
template <bool NT> __device__ __forceinline__ void n2_st2(u64 *p, ulonglong2 v) { // 16 bytes
#ifndef TROYHIP_CPU_EMUL
    if (NT) {
        typedef u64 v2 __attribute__((ext_vector_type(2)));
        v2 w;
        w.x = v.x;
        w.y = v.y;
        __builtin_nontemporal_store(w, reinterpret_cast<v2 *>(p));
        return;
    }
#endif
    *reinterpret_cast<ulonglong2 *>(p) = v;
}
template <bool NT> __device__ __forceinline__ void n2_st(u64 *p, u64 v) {
#ifndef TROYHIP_CPU_EMUL
    if (NT) { __builtin_nontemporal_store(v, p); return; }
#endif
    *p = v;
}
__device__ __forceinline__ unsigned n2_opaque(unsigned v) {
#ifndef TROYHIP_CPU_EMUL
    asm volatile("" : "+v"(v));
#endif
    return v;
}
__device__ __forceinline__ unsigned swz(unsigned f) { return f ^ ((f >> 3) & 7u) ^ (((f >> 6) & 3u) << 3); }

// global coefficient index of flattened tile index f
template <int STRIDED, int NS, int LOGC> __device__ __forceinline__ unsigned g_index(unsigned tile, unsigned f, int logn) {
    if (STRIDED) return ((f >> LOGC) << (logn - NS)) + (tile << LOGC) + (f & ((1u << LOGC) - 1));
    return (tile << N2_LOGT) + f;
}

// One round = R consecutive stages on G = 8 >> R groups of 2^R register-resident points per thread.
//   forward: local stages LS .. LS+R-1 (gaps shrink);  inverse: local stages LS .. LS+R-1 (gaps grow)
__device__ __forceinline__ Shoup to_sgpr(const Shoup w) {
#ifdef TROYHIP_CPU_EMUL
    return w;
#else
    Shoup r;
    r.op = mk64(__builtin_amdgcn_readfirstlane(lo32(w.op)), __builtin_amdgcn_readfirstlane(hi32(w.op)));
    r.quo = mk64(__builtin_amdgcn_readfirstlane(lo32(w.quo)), __builtin_amdgcn_readfirstlane(hi32(w.quo)));
    return r;
#endif
}

template <int INV, int STRIDED, int NS, int LOGC, int LS, int R, bool HOISTP = true> struct Round {
    static constexpr int G = 8 >> R;                       // groups per thread
    static constexpr int THREADS = STRIDED ? (1 << (NS + LOGC - 3)) : N2_THREADS; // a strided pass owns 2^(NS + LOGC) points: 2048 (256 threads) or, wide, 4096 (512)
    static constexpr int NTW = (1 << R) - 1;               // twiddles per group
    static constexpr int LOGPF = NS + LOGC;                // log2 flattened size of one sub-transform
    // forward: first stage gap 2^LOGG, point distance 2^LOGD
    static constexpr int LOGG = INV ? 0 : LOGPF - LS - 1;
    static constexpr int LOGD = INV ? (LS + LOGC) : (LOGG - (R - 1));
    static constexpr bool UNIFORM = LOGD >= 6;             // all lanes of a wave share the twiddles
    static constexpr bool LAST = INV ? (LS + R == NS) : (LS + R == NS);
    // twiddles that live in SGPRs (wave-uniform) or belong to the last round are loaded once per workgroup; the
    // few-distinct-values rounds in between are re-read from L1 per row, which frees ~28 VGPRs (one more wave per SIMD)
    static constexpr bool HOIST = HOISTP; // false: twiddles are re-read (L1/L2) for every row instead of living in 28 VGPRs (the key-switch fused kernel needs them)

    __device__ static __forceinline__ unsigned base_of(unsigned q) {
        const unsigned hi = q >> LOGD, lo = q & ((1u << LOGD) - 1);
        return INV ? ((hi << (LOGD + R)) + lo) : ((hi << (LOGD + R)) + lo); // blocks of 2^(LOGD+R) points either way
    }
    __device__ static __forceinline__ unsigned elem(unsigned q, int e) { return base_of(q) + ((unsigned)e << LOGD); }

    // twiddles of group u of this thread
    __device__ static __forceinline__ void load_tw(Shoup (&tw)[G][NTW], const PrimeDesc &pd, unsigned tile, int logn, int s_first) {
#pragma unroll
        for (int u = 0; u < G; u++) {
            const unsigned q = threadIdx.x + THREADS * u;
            unsigned j0 = g_index<STRIDED, NS, LOGC>(tile, base_of(q), logn);
            if (!INV) {
                const int s = s_first + LS;
                unsigned idx1 = (1u << s) + (j0 >> (logn - s));
                if (UNIFORM) idx1 = __builtin_amdgcn_readfirstlane(idx1);
#pragma unroll
                for (int st = 0; st < R; st++)
#pragma unroll
                    for (int blk = 0; blk < (1 << st); blk++) {
                        const Shoup w = pd.root[(idx1 << st) + blk];
                        tw[u][(1 << st) - 1 + blk] = UNIFORM ? to_sgpr(w) : w;
                    }
            } else {
                const int s = s_first - LS; // global stage of the first (smallest gap) stage of the round
                unsigned blk0 = j0 >> (logn - s);
                if (UNIFORM) blk0 = __builtin_amdgcn_readfirstlane(blk0);
                const unsigned n = 1u << logn;
                int pos = 0;
#pragma unroll
                for (int st = 0; st < R; st++) {
                    const int cs = s - st;
                    const unsigned tbase = n - (2u << cs) + 1 + (blk0 >> st);
#pragma unroll
                    for (int blk = 0; blk < ((1 << R) >> (st + 1)); blk++) {
                        // the very last inverse stage uses the N^-1 pre-scaled twiddle
                        const Shoup w = (STRIDED && (LS + st == NS - 1)) ? pd.iroot_last_scaled : pd.iroot[tbase + blk];
                        tw[u][pos++] = UNIFORM ? to_sgpr(w) : w;
                    }
                }
            }
        }
    }
    // Twiddles that were loaded once per workgroup are CONSUMED here, ahead of the row loop: otherwise the compiler places the wait for them at
    // their first use inside the loop body -- an s_waitcnt vmcnt(0) that every iteration executes, and that then also waits for whatever the
    // iteration has in flight at that point (the next row's staging loads, the key words).  Measured neutral on the headline (8992 / 8965
    // vs 8982 / 8979 ops/s on one box: the staged row has usually landed by then); kept so that the loop body waits where the source says.
    // Hand-written vmcnt(4) for the key words (asm loads) and asm staging loads were tried on top: neutral to -0.4 %, not kept.
    __device__ static __forceinline__ void settle_tw(const Shoup (&tw)[G][NTW]) {
#ifndef TROYHIP_CPU_EMUL
        if (!UNIFORM) {
#pragma unroll
            for (int u = 0; u < G; u++)
#pragma unroll
                for (int i = 0; i < NTW; i++) asm volatile("" ::"v"(tw[u][i].op), "v"(tw[u][i].quo));
        }
#else
        (void)tw;
#endif
    }
    // every stage = exactly four independent butterflies per thread -> one ct_bfly4 / gs_bfly4 call
    // lean (forward only, wave-uniform): the prime is below 2^58 -- guard-free butterflies, every stage adds at most 3p to the
    // value bound (8p at the input of the first pass + 3p * 17 stages at most = 59p < 2^64); the caller reduces with barrett64
    // FP64 form (fpmod.h): x holds the bit patterns of doubles, tw those of (w, w / p); forward stages only
    __device__ static __forceinline__ void compute_fp(u64 (&x)[8], const Shoup (&tw)[G][NTW], const FpPrime &fc, const Shoup inv_n = Shoup{0, 0}) {
        if (N2_EXP & 4) { // removal probe (tools/ntt_probe.sh): no butterflies
#pragma unroll
            for (int u = 0; u < G; u++) x[u] ^= tw[u][0].op;
            return;
        }
        if constexpr (INV) { // Gentleman-Sande: X' = X + Y, Y' = (X - Y) w; the last stage of the transform multiplies both outputs (N^-1 folded in)
#pragma unroll
            for (int st = 0; st < R; st++) {
                const int dist = 1 << st, off = (1 << R) - ((1 << R) >> st);
                const bool last = STRIDED && (LS + st == NS - 1);
#pragma unroll
                for (int u = 0; u < G; u++)
#pragma unroll
                    for (int blk = 0; blk < ((1 << R) >> (st + 1)); blk++)
#pragma unroll
                        for (int k = 0; k < dist; k++) {
                            const int ix = (u << R) + blk * 2 * dist + k, iy = ix + dist;
                            const Shoup &w = tw[u][off + blk];
                            const double X = fp_of_bits(x[ix]), Y = fp_of_bits(x[iy]);
                            const double sum = X + Y, dif = X - Y;
                            x[ix] = fp_bits(last ? fp_mulmod_wp(sum, fp_of_bits(inv_n.op), fp_of_bits(inv_n.quo), fc) : sum);
                            x[iy] = fp_bits(fp_mulmod_wp(dif, fp_of_bits(w.op), fp_of_bits(w.quo), fc));
                        }
            }
            return;
        }
#pragma unroll
        for (int st = 0; st < R; st++) {
#pragma unroll
            for (int u = 0; u < G; u++) {
                const int half = (1 << R) >> (st + 1);
#pragma unroll
                for (int blk = 0; blk < (1 << st); blk++)
#pragma unroll
                    for (int k = 0; k < half; k++) {
                        const int ix = (u << R) + blk * 2 * half + k, iy = ix + half;
                        const Shoup &w = tw[u][(1 << st) - 1 + blk];
                        const double X = fp_of_bits(x[ix]);
                        const double v = fp_mulmod_wp(fp_of_bits(x[iy]), fp_of_bits(w.op), fp_of_bits(w.quo), fc);
                        x[ix] = fp_bits(X + v);
                        x[iy] = fp_bits(X - v);
                    }
            }
        }
    }
    __device__ static __forceinline__ void compute(u64 (&x)[8], const Shoup (&tw)[G][NTW], const PrimeDesc &pd, const bool lean = false) {
        if (N2_EXP & 4) {
#pragma unroll
            for (int u = 0; u < G; u++) x[u] ^= tw[u][0].op;
            return;
        }
        const PrimeConst pc = make_prime_const(pd.p);
#pragma unroll
        for (int st = 0; st < R; st++) {
            u64 X[4], Y[4];
            Shoup w[4];
            int ix[4], iy[4];
            int n = 0;
#pragma unroll
            for (int u = 0; u < G; u++) {
                if (!INV) {
                    const int half = (1 << R) >> (st + 1);
#pragma unroll
                    for (int blk = 0; blk < (1 << st); blk++)
#pragma unroll
                        for (int k = 0; k < half; k++) {
                            ix[n] = (u << R) + blk * 2 * half + k;
                            iy[n] = ix[n] + half;
                            w[n] = tw[u][(1 << st) - 1 + blk];
                            n++;
                        }
                } else {
                    const int dist = 1 << st;
                    // twiddles of stage st start after those of the earlier stages: sum_{t<st} 2^(R-1-t)
                    const int off = (1 << R) - ((1 << R) >> st);
#pragma unroll
                    for (int blk = 0; blk < ((1 << R) >> (st + 1)); blk++)
#pragma unroll
                        for (int k = 0; k < dist; k++) {
                            ix[n] = (u << R) + blk * 2 * dist + k;
                            iy[n] = ix[n] + dist;
                            w[n] = tw[u][off + blk];
                            n++;
                        }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) { X[i] = x[ix[i]]; Y[i] = x[iy[i]]; }
            if (!INV) {
                if (lean) ct_bfly4_ng<UNIFORM>(X, Y, w, pc); else ct_bfly4<UNIFORM>(X, Y, w, pc);
            } else if (STRIDED && (LS + st == NS - 1)) gs_bfly4_last<UNIFORM>(X, Y, w, pd.inv_n, pc);
            else gs_bfly4<UNIFORM>(X, Y, w, pc);
#pragma unroll
            for (int i = 0; i < 4; i++) { x[ix[i]] = X[i]; x[iy[i]] = Y[i]; }
        }
    }
    // t = threadIdx.x, or an opaque copy of it made inside the row loop: the addresses are then formed next to the access instead of
    // being hoisted out of the loop, where eight of them per exchange would live (or be spilled) across the whole kernel
    __device__ static __forceinline__ void lds_read(u64 (&x)[8], const u64 *lds, const unsigned t) {
        if (N2_EXP & 2) return;
#pragma unroll
        for (int u = 0; u < G; u++)
#pragma unroll
            for (int e = 0; e < (1 << R); e++) x[(u << R) + e] = lds[swz(elem(t + THREADS * u, e))];
    }
    __device__ static __forceinline__ void lds_write(const u64 (&x)[8], u64 *lds, const unsigned t) {
        if (N2_EXP & 2) return;
#pragma unroll
        for (int u = 0; u < G; u++)
#pragma unroll
            for (int e = 0; e < (1 << R); e++) lds[swz(elem(t + THREADS * u, e))] = x[(u << R) + e];
    }
    // ---- LDS-DMA staging of the first round's input (contiguous passes): the next row streams into a wave-private
    // 4 KiB staging area while the current row is being transformed, without holding VGPRs for it.  Every DMA
    // instruction moves one CONTIGUOUS 1 KiB chunk (full 128-byte lines); the DMA destination is lane-linear, so any
    // permutation is applied to which 16-byte unit of the chunk a lane fetches:
    // forward (LOGD = 6): the wave reads lane + 64 e -> identity, staging is the sub-transform in natural order;
    // inverse (LOGD = 0): thread t owns units 4t..4t+3 (chunk t/16); lane 16 j + a fetches unit 4a + j, so unit 4a + j
    // sits in slot 16 j + a and the four ds_read_b128 of a thread hit slots (t % 16) + 16 j: the 16 lanes of every
    // ds_read_b128 lane group see 16 different slots mod 16 (conflict-free).
    __device__ static __forceinline__ unsigned unit_perm(unsigned lane) { return 4 * (lane & 15) + (lane >> 4); }
    __device__ static __forceinline__ void stage_issue(const u64 *row, unsigned tile, u64 *wave_stage) {
        static_assert(!STRIDED && NS == 9 && R == 3 && LS == 0, "staging is defined for the first round of the contiguous pass");
        const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const u64 *src = row + ((size_t)tile << N2_LOGT) + 512 * w + 2 * (INV ? unit_perm(lane) : lane);
#pragma unroll
        for (int i = 0; i < 4; i++) TROY_GLDS16_POL(src + 128 * i, wave_stage + 128 * i, N2_DMA_POL);
    }
    __device__ static __forceinline__ void stage_read(u64 (&x)[8], const u64 *wave_stage) {
        const unsigned lane = threadIdx.x & 63;
        if (INV) {
            const u64 *mine = wave_stage + 128 * (lane >> 4) + 2 * (lane & 15);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(mine + 32 * j);
                x[2 * j] = v.x;
                x[2 * j + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = wave_stage[lane + 64 * e];
        }
    }
    // Last round of the forward contiguous pass (LOGD = 0): thread t holds 64 contiguous bytes.  Written directly that
    // is four stores of 16 bytes every 64 bytes (each touching all 32 lines of the wave's 4 KiB); instead the wave
    // transposes through its private exchange area with the same unit permutation and issues four stores of one
    // contiguous 1 KiB each.
    template <bool NT = (N2_NT & 8) != 0> __device__ static __forceinline__ void store_via_lds(const u64 (&x)[8], u64 *row, unsigned tile, u64 *wave_xchg) {
        const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        u64 *mine = wave_xchg + 128 * (lane >> 4) + 2 * (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            ulonglong2 v;
            v.x = x[2 * j];
            v.y = x[2 * j + 1];
            *reinterpret_cast<ulonglong2 *>(mine + 32 * j) = v;
        }
        TROY_WAVE_SYNC();
        u64 *dst = row + ((size_t)tile << N2_LOGT) + 512 * w + 2 * unit_perm(lane);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(wave_xchg + 128 * c + 2 * lane);
            n2_st2<NT>(dst + 128 * c, v);
        }
    }
    // MAC = 2 epilogue of the forward contiguous pass: virtual row vr = mm % 4 of (a0, a1, b0, b1); x is the lazy transform
    // ([0, 8p)) of this thread's 8 consecutive coefficients.  Rows 0..2 are parked canonical in tx; row 3 completes the
    // tensor (evaluator.cpp:626-702: every product reduced, the two middle products added modulo p).
    __device__ static __forceinline__ void tensor_epilogue(u64 (&x)[8], u64 (&tx)[3][8], unsigned mm, u64 *out, unsigned period, unsigned slot, unsigned tile, int logn,
                                                           const Mod &m, u64 *xchg, const bool lean, const FpPrime *fc = nullptr) {
        const PrimeConst pc = make_prime_const(m.p);
        if (fc) {
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = fp_canonical(fp_of_bits(x[e]), *fc, m.p);
        } else if (lean) { // guard-free transform: values below 59p.  The products do not need canonical factors: below 4p each (lite_reduce4: 4 instructions
            // per value against 10 for the canonical residue) a product is below 16 p^2 and the middle sum below 32 p^2 < 2^(2k+5), k = bit length of p <= 58;
            // reduce_prod's quotient estimate floor((x >> (k-1)) mu / 2^64) keeps its error of at most 2 as long as x >> (k-1) < 2^64, i.e. up to 2^(k+63)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                u64 v[4] = {x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3]};
                lite_reduce4(v, (u32)m.cr1, pc);
#pragma unroll
                for (int i = 0; i < 4; i++) x[4 * h + i] = v[i];
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                u64 v[4] = {x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3]};
                reduce4_from_8p(v, pc);
#pragma unroll
                for (int i = 0; i < 4; i++) x[4 * h + i] = v[i];
            }
        }
        const unsigned vr = mm & 3;
        if (vr < 3) {
#pragma unroll
            for (int i = 0; i < 3; i++)
                if ((int)vr == i) {
#pragma unroll
                    for (int e = 0; e < 8; e++) tx[i][e] = x[e];
                }
            return;
        }
        // the four operands are canonical (guarded / FP64 transforms) or below 4p (guard-free): every product and the middle sum are within what
        // reduce_prod takes -- (a0 b1 + a1 b0) mod p is the same residue as the reference's add of the two reduced products
        const unsigned b = mm >> 2;
        const ProdMod pm = make_prod_mod(m);
        u64 d[3][8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const u64 a0 = tx[0][e], a1 = tx[1][e], b0 = tx[2][e], b1 = x[e];
            d[0][e] = reduce_prod((u128)a0 * b0, pm);
            d[1][e] = reduce_prod((u128)a0 * b1 + (u128)a1 * b0, pm);
            d[2][e] = reduce_prod((u128)a1 * b1, pm);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            u64 *orow = out + ((((u64)b * 3 + i) * period + slot) << logn);
            if (N2_COALESCED_STORE && xchg) { // 1 KiB-contiguous stores through the wave's exchange area, as the plain pass
                TROY_WAVE_SYNC();
                store_via_lds(d[i], orow, tile, xchg + 512 * (threadIdx.x >> 6));
                continue;
            }
            ulonglong2 *op = reinterpret_cast<ulonglong2 *>(orow + ((u64)tile << N2_LOGT) + 8 * threadIdx.x);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                ulonglong2 v;
                v.x = d[i][2 * e];
                v.y = d[i][2 * e + 1];
                op[e] = v;
            }
        }
    }
    // global access; consecutive-element runs are moved 16 bytes at a time
    template <int REDUCE> __device__ static __forceinline__ void g_read(u64 (&x)[8], const u64 *row, unsigned tile, int logn, const Mod &m, const unsigned t = threadIdx.x) {
        if (N2_EXP & 1) {
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = (u64)(uintptr_t)row + t * 8 + e;
            return;
        }
#pragma unroll
        for (int u = 0; u < G; u++) {
            const unsigned q = t + THREADS * u;
            if (LOGD == 0) {
#pragma unroll
                for (int e = 0; e < (1 << R); e += 2) {
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(row + g_index<STRIDED, NS, LOGC>(tile, elem(q, e), logn));
                    x[(u << R) + e] = v.x;
                    x[(u << R) + e + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int e = 0; e < (1 << R); e++) x[(u << R) + e] = n2_ld<STRIDED && !INV && (N2_NT & 2) != 0>(row + g_index<STRIDED, NS, LOGC>(tile, elem(q, e), logn));
            }
        }
        (void)m;
    }
    // FINAL: 0 keep lazy range, 1 forward final ([0,8p) -> [0,p)), 2 inverse final ([0,4p) -> [0,p))
    template <int FINAL> __device__ static __forceinline__ void g_write(u64 (&x)[8], u64 *row, unsigned tile, int logn, const Mod &m, const bool lean, u64 *xchg = nullptr,
                                                                        const FpPrime *fc = nullptr) {
        const u64 p = m.p;
        if (FINAL && fc) { // FP64 instance: the doubles become canonical words, then the common store code (which skips its own reduction)
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = fp_canonical(fp_of_bits(x[e]), *fc, p);
        }
        if (N2_EXP & 1) {
            u64 acc = 0;
#pragma unroll
            for (int e = 0; e < 8; e++) acc ^= x[e];
            if (acc == 0x123456789abcdefull) row[threadIdx.x] = acc; // never true in practice; keeps the work alive
            return;
        }
        if (FINAL && fc) {
        } else if (FINAL == 1 && lean) { // guard-free forward transform: values below 59p
            lean_final<8>(x, make_lean_final(m.p, m.cr1), make_prime_const(p));
        } else if (FINAL) {
            const PrimeConst pc = make_prime_const(p);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                u64 v[4] = {x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3]};
                if (FINAL == 1) reduce4_from_8p(v, pc); else reduce4_from_4p(v, pc);
#pragma unroll
                for (int i = 0; i < 4; i++) x[4 * h + i] = v[i];
            }
        }
        if constexpr (N2_COALESCED_STORE && !STRIDED && !INV && NS == 9 && R == 3 && LOGD == 0) {
            if (xchg) {
                TROY_WAVE_SYNC(); // the exchange area was last read by this round's lds_read
                store_via_lds(x, row, tile, xchg + 512 * (threadIdx.x >> 6));
                return;
            }
        }
#pragma unroll
        for (int u = 0; u < G; u++) {
            const unsigned q = threadIdx.x + THREADS * u;
            if (LOGD == 0) {
#pragma unroll
                for (int e = 0; e < (1 << R); e += 2) {
                    ulonglong2 v;
                    v.x = x[(u << R) + e];
                    v.y = x[(u << R) + e + 1];
                    *reinterpret_cast<ulonglong2 *>(row + g_index<STRIDED, NS, LOGC>(tile, elem(q, e), logn)) = v;
                }
            } else {
#pragma unroll
                for (int e = 0; e < (1 << R); e++) n2_st<STRIDED && ((!INV && (N2_NT & 1) != 0) || (INV && (N2_NT & 4) != 0))>(row + g_index<STRIDED, NS, LOGC>(tile, elem(q, e), logn), x[(u << R) + e]);
            }
        }
    }
    // FINAL = 3 / 4: the mod-down of BFV / BGV key switching (evaluator.cpp:2528-2648, the element-wise ks_moddown_kernel of poly.hip) instead
    // of the store.  The row is acc[o][slot] transformed with a prime table whose N^-1 constants carry qk^-1 (Context::d_desc_md), so x is
    // acc_j qk^-1, lazily below 4p; the special limb acc[o][dl] is already in coefficient form.  With al = acc[o][dl][n]:
    //   BFV: ct[b][k][slot][n] += x + ([half]_p - [(al + half) mod qk]_p) qk^-1
    //   BGV: ct[b][k][slot][n] += x - [al + k_t qk]_p qk^-1,   k_t = -al qk^-1 mod t; the integer al + k_t qk comes from md_share
    // every term canonical before the final addition, as in the element-wise kernel: same residues, same stored values.
    template <int FINAL> __device__ static __forceinline__ void md_write(u64 (&x)[8], const Ntt2Args &a, unsigned o, unsigned slot, unsigned tile, int logn, const Mod &m,
                                                                         const PrimeDesc &pd, const FpPrime *fc = nullptr) {
        static_assert(INV && STRIDED && G == 1, "the mod-down epilogue belongs to the last inverse pass");
        const u64 p = m.p;
        const PrimeConst pc = make_prime_const(p);
        const u64 *sp = a.data + (((u64)o * a.map.period + a.md_dl) << logn);
        u64 *ct = a.md_ct + (u64)(o >> 1) * a.md_ct_bstride + (((u64)(o & 1) * a.md_dl + slot) << logn);
        const u64 *base = a.md_base ? a.md_base + (u64)(o >> 1) * a.md_base_bstride + (((u64)(o & 1) * a.md_dl + slot) << logn) : ct;
        const bool from_zero = a.md_base && (o & 1) && a.md_base_polys < 2;
        const ulonglong2 *sh = reinterpret_cast<const ulonglong2 *>(a.md_share) + ((u64)o << logn);
        const u64 c_p = FINAL == 3 ? barrett64(a.md_half, m) : 0; // [half]_p
#pragma unroll
        for (int h = 0; h < 2; h++) {
            u64 v[4] = {x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3]};
            if (fc) {
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = fp_canonical(fp_of_bits(v[i]), *fc, p);
            } else {
                reduce4_from_4p(v, pc);
            }
            u64 al[4], ah[4], c[4];
            unsigned n[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                n[i] = g_index<STRIDED, NS, LOGC>(tile, elem(threadIdx.x, 4 * h + i), logn);
                if (FINAL == 3) { al[i] = sp[n[i]]; ah[i] = 0; }
                else { const ulonglong2 t = sh[n[i]]; al[i] = t.x; ah[i] = t.y; }
                c[i] = from_zero ? 0 : base[n[i]];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                u64 w; // minus the special limb's share before the multiplication by qk^-1, in (0, 2p]
                if (FINAL == 3) {
                    u64 tl = al[i] + a.md_half;
                    tl = tl >= a.md_qk ? tl - a.md_qk : tl;
                    w = p + c_p - barrett64(tl, m);
                } else {
                    w = p - barrett128(al[i], ah[i], m);
                }
                u64 r = v[i] + mul_shoup(w, pd.aux, p);
                r = r >= p ? r - p : r;
                ct[n[i]] = addmod(c[i], r, p);
            }
        }
    }
};

// stage split of a pass into rounds: up to four rounds R0..R3 (0 = unused)
template <int NS> struct Plan;
template <> struct Plan<3> { static constexpr int r[4] = {3, 0, 0, 0}; };
template <> struct Plan<4> { static constexpr int r[4] = {3, 1, 0, 0}; };
template <> struct Plan<5> { static constexpr int r[4] = {3, 2, 0, 0}; };
template <> struct Plan<6> { static constexpr int r[4] = {3, 3, 0, 0}; };
template <> struct Plan<7> { static constexpr int r[4] = {3, 1, 3, 0}; }; // the single stage in the MIDDLE: wave-uniform twiddles (SGPRs) in both directions; {3, 3, 1} spilled
template <> struct Plan<9> { static constexpr int r[4] = {3, 3, 3, 0}; };
template <> struct Plan<10> { static constexpr int r[4] = {3, 3, 3, 1}; };
template <> struct Plan<11> { static constexpr int r[4] = {3, 3, 3, 2}; };

// FP = 1: the FP64 instances (ntt2_fp_kernel below; fpmod.h) -- x then holds the bit patterns of doubles (exact integers, signed lazy range), the
// twiddles are (w, w / p) from PrimeDesc::root_fp, every data movement (HBM, LDS, LDS-DMA) is the same 64-bit traffic as in the integer form
template <int INV, int STRIDED, int NS, int LOGC, int FINAL, int REDUCE, int MAC, int FP>
__device__ __forceinline__ void ntt2_body(const Ntt2Args &a) {
    static_assert(!FP || !(INV && MAC), "FP64 instances: every pass of the two-pass transform and its fusions");
    // [0] the exchange buffer of the rounds, [1] the LDS-DMA staging area of the contiguous passes (one row ahead).  Two rows ahead in the key-switch passes
    // (a third 16 KiB buffer, s_waitcnt vmcnt(4) for the older of two rows in flight) was measured in round 4: 1-2 % SLOWER on all four workloads --
    // the wait at the top of a row is not what those passes are bound by
    constexpr int LOGT = STRIDED ? NS + LOGC : N2_LOGT; // points of the workgroup's tile: N2_T, or 4096 for the wide strided passes (512 threads, 64 KiB of LDS)
    __shared__ u64 lds[2][1 << LOGT];
    using P = Plan<NS>;
    // NS == 9, contiguous: every 512-point sub-transform is owned by ONE wave in every round (thread t's points never
    // leave [512*(t/64), 512*(t/64)+512), and the XOR swizzle only permutes address bits 0..4), so the LDS exchange
    // needs no workgroup barrier at all: the four waves run fully decoupled and overlap each other's HBM phases.
    constexpr bool WAVE_PRIVATE = !STRIDED && NS == 9;
    // MAC = 1: key-switch inner product (BFV / BGV); MAC = 3: the same for CKKS, where the row of digit k == output slot is the NTT-form input
    // itself and is neither staged nor transformed; REDUCE = 2: the digit-reducing first pass of that key switch, which does not expand those rows
    constexpr bool KS = MAC == 1 || MAC == 3;
    // Which LDS exchanges form their addresses per row instead of once per workgroup (bit 0: round 0 write, 1: round 1 read, 2: round 1
    // write, 3: round 2 read, 4: round 2 write, 5: round 3 read).  Hoisted, the eight swizzled addresses of an exchange live across the
    // whole row loop; in the instances at the 128-register limit the compiler spilled them, and a spill reload waits with vmcnt(0) --
    // i.e. for the NEXT row's loads, which were issued before it: the HBM latency landed in the middle of every row (forward
    // contiguous pass 2038 -> 1525 us per 6720 rows of N = 2^16; three XORs per address are nothing next to that).
    constexpr int FRESH = (!STRIDED && !INV && NS == 9 && MAC == 0) ? (FINAL ? 15 : 1)            // plain forward contiguous pass (it spilled)
                          : (STRIDED && !INV && NS == 7)                 ? 1                         // 7-stage strided forward pass (it spilled; more than the first exchange costs 2-6 % there)
                          : (!STRIDED && NS == 10)                       ? 63                        // N = 2^17
                          : (FINAL >= 3 && NS >= 6)                      ? N2_FRESH_MD               // mod-down epilogues: room for their operands
                          : MAC == 2                                     ? N2_FRESH_TENSOR           // tensor pass: 168 VGPRs = 3 waves per SIMD
                          : (FP && MAC)                                  ? N2_FRESH_FP_MAC           // FP64 key-switch pass: 184 -> 168 VGPRs = 3 waves per SIMD
                          : (STRIDED && !INV && !REDUCE && NS >= 4 && NS <= 6) ? 15                  // plain strided forward pass: room for the register prefetch (PF)
                                                                         : 0;
    auto round_sync = [&]() {
        if (N2_EXP & 2) return;
        if (WAVE_PRIVATE) TROY_WAVE_SYNC(); else __syncthreads();
    };
    // inverse passes run the same round list but with growing gaps, so their local-stage offsets are the same sums
    constexpr int R0 = P::r[0], R1 = P::r[1], R2 = P::r[2], R3 = P::r[3];
    constexpr int NR = (R0 > 0) + (R1 > 0) + (R2 > 0) + (R3 > 0);
    // forward executes rounds 0..NR-1 with shrinking gaps; inverse executes the reversed list (small rounds first keeps
    // the radix-8 rounds on the large gaps, i.e. on the coalescing-friendly side)
    constexpr int Q0 = INV ? P::r[NR - 1] : R0;
    constexpr int Q1 = NR > 1 ? (INV ? P::r[NR - 2] : R1) : 0;
    constexpr int Q2 = NR > 2 ? (INV ? P::r[NR - 3] : R2) : 0;
    constexpr int Q3 = NR > 3 ? (INV ? P::r[NR - 4] : R3) : 0;
    using Rd0 = Round<INV, STRIDED, NS, LOGC, 0, Q0>;
    using Rd1 = Round<INV, STRIDED, NS, LOGC, Q0, Q1 ? Q1 : 1, MAC != 2 || N2_TENSOR_HOIST1>;
    using Rd2 = Round<INV, STRIDED, NS, LOGC, Q0 + Q1, Q2 ? Q2 : 1, (MAC == 2 ? N2_TENSOR_HOIST2 : (MAC != 1 && MAC != 3) || N2_MAC_HOIST)>;
    using Rd3 = Round<INV, STRIDED, NS, LOGC, Q0 + Q1 + Q2, Q3 ? Q3 : 1>;

    const unsigned tile = blockIdx.x & ((1u << a.tiles_per_row_log) - 1);
    const unsigned grp = blockIdx.x >> a.tiles_per_row_log;
    // group order: prime slot major (workgroups in flight share the twiddles, and in the fused key-switch pass the key window),
    // or slot FASTEST (the first key-switch pass: the L+1 readers of one source digit run together and hit in L2)
    const unsigned sidx = a.slot_fastest ? grp % a.nsel : grp / a.chunks, chunk = a.slot_fastest ? grp / a.nsel : grp % a.chunks;
    const unsigned slot = (unsigned)__builtin_amdgcn_readfirstlane((int)a.sel[sidx]);
    PrimeDesc pd = a.primes[a.map.id[slot]];
    if constexpr (FP) { pd.root = pd.root_fp; pd.iroot = pd.iroot_fp; pd.inv_n = pd.inv_n_fp; pd.iroot_last_scaled = pd.iroot_last_scaled_fp; }
    const FpPrime fc_ = make_fp_prime_uniform(FP ? pd.p : 1);
    const FpPrime &fc = fc_;
    const FpPrime *const fcp = FP ? &fc_ : nullptr;
    const Mod m = mod_of(pd);
    const bool need_reduce = !FP && REDUCE && (a.src_bound == 0 || (pd.p >> 61) != 0 || a.src_bound > 8 * pd.p);
#ifdef N2_CENSUS_LEAN // tools/isa_census.py --flags=-DN2_CENSUS_LEAN: the guard-free path alone, so that a loop's count is one path's count (never shipped)
    const bool lean = !INV;
#else
    const bool lean = !INV && ((a.map.lean >> slot) & 1); // wave-uniform: prime below 2^58 -> guard-free forward butterflies (bfly.h)
#endif
    const int logn = a.logn;
    const int k1 = STRIDED ? NS : logn - NS;
    const int s_first = INV ? (STRIDED ? k1 - 1 : logn - 1) : (STRIDED ? 0 : k1);

    Shoup tw0[Rd0::G][Rd0::NTW], tw1[Rd1::G][Rd1::NTW], tw2[Rd2::G][Rd2::NTW], tw3[Rd3::G][Rd3::NTW];
    if constexpr (Rd0::HOIST) Rd0::load_tw(tw0, pd, tile, logn, s_first);
    if constexpr (NR > 1 && Rd1::HOIST) Rd1::load_tw(tw1, pd, tile, logn, s_first);
    if constexpr (NR > 2 && Rd2::HOIST) Rd2::load_tw(tw2, pd, tile, logn, s_first);
    if constexpr (NR > 3 && Rd3::HOIST) Rd3::load_tw(tw3, pd, tile, logn, s_first);
    if constexpr (Rd0::HOIST) Rd0::settle_tw(tw0);
    if constexpr (NR > 1 && Rd1::HOIST) Rd1::settle_tw(tw1);
    if constexpr (NR > 2 && Rd2::HOIST) Rd2::settle_tw(tw2);
    if constexpr (NR > 3 && Rd3::HOIST) Rd3::settle_tw(tw3);

    const unsigned m_begin = chunk * a.rows_per_wg;
    const unsigned m_end = (m_begin + a.rows_per_wg < a.m_total) ? m_begin + a.rows_per_wg : a.m_total;
    const unsigned inner = a.map.inner, period = a.map.period;
    // FP64 instances: reduce the eight values before round r where the host's bound walk says so (fp_plan, fpmod.h; wave-uniform)
    auto fp_guard = [&](u64 (&v)[8], int r) {
        if constexpr (FP) {
            if ((a.fp_red_mask >> r) & 1) {
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = fp_bits(fp_reduce(fp_of_bits(v[e]), fc));
            }
        }
    };
    // software pipeline over the rows of this group: the loads of row mm+1 are in flight while row mm is transformed
    // row mm = o * inner + k: (o, k) is carried along the row loop (one division per workgroup, not three per row)
    auto row_ptrs = [&](unsigned mm, unsigned o, unsigned k, u64 *&row, const u64 *&in) {
        if constexpr (MAC == 2) { // virtual row mm = 4 b + vr: vr 0,1 -> a0, a1 ; 2,3 -> b0, b1
            const unsigned b = mm >> 2, vr = mm & 3;
            row = (vr < 2 ? a.data : a.tensor_b) + ((((u64)b * 2 + (vr & 1)) * period + slot) << logn);
            in = row;
            return;
        }
        const u64 r = ((u64)o * period + slot) * inner + k;
        row = a.data + (r << logn);
        in = (REDUCE || a.src) ? (a.src_same_layout ? a.src + (r << logn) : a.src + (u64)o * a.src_ostride + ((u64)k << logn)) : row;
        if constexpr (!REDUCE && STRIDED && !INV) {
            if (a.src_same_layout == 2) // wave-uniform: which buffer this row's prime slot lives in
                in = slot >= a.src_slots ? row : (o < a.src_split ? a.src + (((u64)o * a.src_slots + slot) << logn) : a.src2 + (((u64)(o - a.src_split) * a.src_slots + slot) << logn));
        }
        if ((N2_EXP & 8) && !STRIDED) in = a.data + ((r & 63) << logn); // probe: the contiguous pass reads a 16 MB window (L2-resident input)
    };
    constexpr bool DMA = N2_DMA && WAVE_PRIVATE && !(N2_EXP & 1);
    static_assert(MAC != 3 || DMA || (N2_EXP & 1), "the CKKS key-switch pass (MAC = 3) takes its rows from the LDS-DMA staging area: build it with N2_DMA = 1 and without the no-HBM probe");
    // plain strided forward passes of two rounds: the next row is requested into a second register set before the last round's butterflies
    // (the registers come from forming the LDS addresses per row, FRESH, so the kernel stays at four waves per SIMD): -2 % on that pass.
    // Not the digit-reducing first pass of key switching, whose L2-resident sources arrive fast enough anyway: +6 % there.
    constexpr bool PF = STRIDED && !INV && NR == 2 && !REDUCE && !(N2_EXP & 1);
    u64 xn[PF ? 8 : 1];
    static_assert(!MAC || (!INV && !STRIDED && NS == 9), "the inner product is fused into the forward contiguous pass");
    u64 tx[MAC == 2 ? 3 : 1][8]; // MAC = 2: the transforms of a0, a1, b0 while b1 is being computed
    Acc128 macc[2][2][4]; // [key component][group of four coefficients][coefficient]
    if (KS) {
#pragma unroll
        for (int cpt = 0; cpt < 2; cpt++)
#pragma unroll
            for (int g = 0; g < 2; g++)
#pragma unroll
                for (int e = 0; e < 4; e++) macc[cpt][g][e] = Acc128{0, 0, 0, 0};
    }
    // FP64 instances: one double per accumulator (the products are reduced modulo p as they are formed: 6 + 1 instructions per term and two
    // for the key word's conversion, against 10 for the 128-bit integer form -- and 32 VGPRs of accumulators instead of 64)
    double facc[2][FP ? 8 : 1];
    if constexpr (FP) {
#pragma unroll
        for (int cpt = 0; cpt < 2; cpt++)
#pragma unroll
            for (int e = 0; e < 8; e++) facc[cpt][e] = 0.0;
    }
    auto mac_row = [&](const u64 (&xr)[8], const ulonglong2 (&kw)[2][KS ? 4 : 1], unsigned row_no) {
        if constexpr (FP && KS) {
#pragma unroll
            for (int cpt = 0; cpt < 2; cpt++)
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const u64 kword = (e & 1) ? kw[cpt][e >> 1].y : kw[cpt][e >> 1].x;
                    // N2_EXP & 16 (removal probe, wrong results): the key word taken as a double as it lies -- what storing the key as doubles would save
                    facc[cpt][e] += fp_mulmod_pinv(fp_of_bits(xr[e]), (N2_EXP & 16) ? fp_of_bits(kword) : fp_from_u64(kword), fc);
                }
            if (a.fp_acc_every && (row_no + 1) % a.fp_acc_every == 0) { // wave-uniform; the sums stay below 2^53 (launch_ntt2_ks_mac)
#pragma unroll
                for (int cpt = 0; cpt < 2; cpt++)
#pragma unroll
                    for (int e = 0; e < 8; e++) facc[cpt][e] = fp_reduce(facc[cpt][e], fc);
            }
        } else if constexpr (KS) {
#pragma unroll
            for (int cpt = 0; cpt < 2; cpt++) {
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    const ulonglong2 k01 = kw[cpt][2 * g], k23 = kw[cpt][2 * g + 1];
                    const u64 kk[4] = {k01.x, k01.y, k23.x, k23.y};
                    const u64 xx[4] = {xr[4 * g], xr[4 * g + 1], xr[4 * g + 2], xr[4 * g + 3]};
                    mac128x4(macc[cpt][g], xx, kk);
                }
            }
        }
        (void)row_no;
    };
    // read once: indexing the argument block with `slot` is a memory load, and inside the row loop it sat, with its wait, in front of the key loads
    const unsigned key_limb = KS ? (unsigned)__builtin_amdgcn_readfirstlane((int)a.mac_key_limb[slot]) : 0;
    // the sums leave through the wave's exchange area, one contiguous KiB per wave and instruction (Rd2::store_via_lds): written from a thread's eight
    // consecutive coefficients they were 16 bytes per lane every 64 bytes -- the access pattern that cost the single-pass forward kernel 9 % (ntt1.hip).
    // Same-box A/B (tools/r4_ab_maclin.sh): accumulating pass -1..-3 % at the headline, -4 % in the CKKS chain, -8 % at configs[1], BGV N = 2^16 unchanged.
    // (The key LOADS in that pattern are harmless: a per-row transposition that made them contiguous as well measured the same or 1-2 % slower.)
    constexpr bool MAC_STORE_LINEAR = KS && N2_MAC_STORE_LINEAR && WAVE_PRIVATE;
    u64 *const wave_stage = lds[1] + 512 * (threadIdx.x >> 6);
    u64 x[8];
    unsigned ro = m_begin / inner, rk = m_begin - ro * inner; // (o, k) of the current row
    unsigned parity = 0;
    // CKKS key switch: the row whose digit index equals the output slot is the NTT-form input itself (evaluator.cpp:2424-2427): the first
    // pass does not expand it and the fused second pass neither stages nor transforms it (one row in L + 1 of both passes)
    auto is_diag = [&](unsigned k) { return (MAC == 3 || REDUCE == 2) && k == slot; };
    {
        u64 *row0; const u64 *in0;
        row_ptrs(m_begin, ro, rk, row0, in0);
        if (!is_diag(rk)) {
            if constexpr (DMA) Rd0::stage_issue(in0, tile, wave_stage);
            else Rd0::template g_read<REDUCE>(x, in0, tile, logn, m);
        }
    }
    for (unsigned mm = m_begin; mm < m_end; mm++) {
        u64 *row; const u64 *in;
        row_ptrs(mm, ro, rk, row, in);
        const unsigned no = rk + 1 == inner ? ro + 1 : ro, nk = rk + 1 == inner ? 0 : rk + 1; // (o, k) of row mm + 1
        const bool next_wanted = mm + 1 < m_end && !is_diag(nk);
        const bool skip_row = REDUCE == 2 && is_diag(rk); // nothing to expand: the body below is skipped as a whole, the software pipeline goes on
        if constexpr (DMA) TROY_WAIT_VMEM(); // this wave's staged row has landed (and its previous stores are out)
        ulonglong2 kv[2][KS ? 4 : 1]; // MAC: this row's key words, requested now -- BEFORE the next row's staging loads, so that the wait for them (vmcnt counts in
        // order) does not include the staging loads' HBM latency -- and used after the three rounds
        auto load_keys = [&]() {
            const unsigned kk = rk;
            const u64 *kp = a.mac_key + ((((u64)kk * 2) * a.mac_K + key_limb) << logn) + ((u64)tile << N2_LOGT) + 8 * threadIdx.x;
#pragma unroll
            for (int cpt = 0; cpt < 2; cpt++) {
                const ulonglong2 *kq = reinterpret_cast<const ulonglong2 *>(kp + ((u64)cpt * a.mac_K << logn));
#pragma unroll
                for (int e = 0; e < 4; e++) kv[cpt][KS ? e : 0] = kq[e];
            }
        };
        if constexpr (KS) load_keys();
        if constexpr (MAC == 3) { // CKKS key switch: one code path for both kinds of rows, so that the accumulate code exists once
            const bool diag = is_diag(rk);
            if (!diag) {
                Rd0::stage_read(x, wave_stage);
                TROY_WAIT_LDS();
            }
            if (next_wanted) {
                u64 *nrow; const u64 *nin;
                row_ptrs(mm + 1, no, nk, nrow, nin);
                Rd0::stage_issue(nin, tile, wave_stage);
            }
            if (!diag) {
                u64 *buf = lds[0];
                fp_guard(x, 0);
                if constexpr (FP) Rd0::compute_fp(x, tw0, fc, pd.inv_n); else Rd0::compute(x, tw0, pd, lean);
                Rd0::lds_write(x, buf, (FRESH & 1) ? n2_opaque(threadIdx.x) : threadIdx.x);
                round_sync();
                Rd1::lds_read(x, buf, (FRESH & 2) ? n2_opaque(threadIdx.x) : threadIdx.x);
                fp_guard(x, 1);
                if constexpr (FP) Rd1::compute_fp(x, tw1, fc, pd.inv_n); else Rd1::compute(x, tw1, pd, lean);
                Rd1::lds_write(x, buf, (FRESH & 4) ? n2_opaque(threadIdx.x) : threadIdx.x);
                round_sync();
                if constexpr (!Rd2::HOIST) Rd2::load_tw(tw2, pd, tile, logn, s_first);
                Rd2::lds_read(x, buf, (FRESH & 8) ? n2_opaque(threadIdx.x) : threadIdx.x);
                fp_guard(x, 2);
                if constexpr (FP) Rd2::compute_fp(x, tw2, fc, pd.inv_n); else Rd2::compute(x, tw2, pd, lean);
                if constexpr (FP) {
                } else if (!a.mac_lazy && lean) {
#pragma unroll
                    for (int e = 0; e < 8; e++) x[e] = barrett64(x[e], m);
                } else if (!a.mac_lazy) {
                    const PrimeConst pc = make_prime_const(pd.p);
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        u64 v[4] = {x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3]};
                        reduce4_from_8p(v, pc);
#pragma unroll
                        for (int i = 0; i < 4; i++) x[4 * h + i] = v[i];
                    }
                }
            } else { // the operand is the NTT-form input limb: no staged row, no transform
                const ulonglong2 *tp = reinterpret_cast<const ulonglong2 *>(a.mac_target + (u64)ro * a.mac_tstride + ((u64)rk << logn) + ((u64)tile << N2_LOGT) + 8 * threadIdx.x);
#pragma unroll
                for (int e = 0; e < 4; e++) { const ulonglong2 v = tp[e]; x[2 * e] = v.x; x[2 * e + 1] = v.y; }
                if constexpr (FP) {
#pragma unroll
                    for (int e = 0; e < 8; e++) x[e] = fp_bits(fp_from_u64(x[e]));
                }
            }
            mac_row(x, kv, mm - m_begin);
            ro = no;
            rk = nk;
            continue;
        }
        if constexpr (DMA) {
            Rd0::stage_read(x, wave_stage);
            TROY_WAIT_LDS();             // ... and has been read before the next row overwrites it
            if (next_wanted) {
                u64 *nrow; const u64 *nin;
                row_ptrs(mm + 1, no, nk, nrow, nin);
                Rd0::stage_issue(nin, tile, wave_stage);
            }
        }
        if (!skip_row) {
        u64 *buf = DMA ? lds[0] : lds[parity]; // the exchange buffers alternate over the rows that are actually processed (skipped rows have no barriers)
        if (REDUCE && need_reduce) { // wave-uniform: the butterflies take any input below 8p (ct_bfly4), most prime sets never need this
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = barrett64(x[e], m);
        }
        if constexpr (!Rd0::HOIST) Rd0::load_tw(tw0, pd, tile, logn, s_first);
        if constexpr (FP && REDUCE) { // the digit's residues become doubles: exact below 2^52; a wide source prime (>= 2^50) is reduced modulo this row's prime first
            if ((a.fp_src_wide >> rk) & 1) {
#pragma unroll
                for (int e = 0; e < 8; e++) x[e] = barrett64(x[e], m);
            }
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = fp_bits(fp_from_u64(x[e]));
        } else if constexpr (FP && ((!INV && STRIDED) || (INV && !STRIDED))) { // first pass of a plain transform: canonical residues of this row's own prime
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = fp_bits(fp_from_u64(x[e]));
        }
        fp_guard(x, 0);
        if constexpr (FP) Rd0::compute_fp(x, tw0, fc, pd.inv_n); else Rd0::compute(x, tw0, pd, lean);
        if constexpr (NR == 1) {
            if constexpr (FINAL >= 3) Rd0::template md_write<FINAL>(x, a, mm, slot, tile, logn, m, pd, fcp);
            else Rd0::template g_write<FINAL>(x, row, tile, logn, m, lean, nullptr, fcp);
        } else {
            Rd0::lds_write(x, buf, (FRESH & 1) ? n2_opaque(threadIdx.x) : threadIdx.x);
            round_sync();
            if constexpr (!Rd1::HOIST) Rd1::load_tw(tw1, pd, tile, logn, s_first);
            Rd1::lds_read(x, buf, (FRESH & 2) ? n2_opaque(threadIdx.x) : threadIdx.x);
            if constexpr (PF) {
                if (mm + 1 < m_end) {
                    u64 *nrow; const u64 *nin;
                    row_ptrs(mm + 1, no, nk, nrow, nin);
                    Rd0::template g_read<REDUCE>(xn, nin, tile, logn, m, n2_opaque(threadIdx.x));
                }
            }
            fp_guard(x, 1);
            if constexpr (FP) Rd1::compute_fp(x, tw1, fc, pd.inv_n); else Rd1::compute(x, tw1, pd, lean);
            if constexpr (NR == 2) {
                if constexpr (FINAL >= 3) Rd1::template md_write<FINAL>(x, a, mm, slot, tile, logn, m, pd, fcp);
                else Rd1::template g_write<FINAL>(x, row, tile, logn, m, lean, nullptr, fcp);
            } else {
                Rd1::lds_write(x, buf, (FRESH & 4) ? n2_opaque(threadIdx.x) : threadIdx.x);
                round_sync();
                if constexpr (!Rd2::HOIST) Rd2::load_tw(tw2, pd, tile, logn, s_first);
                Rd2::lds_read(x, buf, (FRESH & 8) ? n2_opaque(threadIdx.x) : threadIdx.x);
                fp_guard(x, 2);
                if constexpr (FP) Rd2::compute_fp(x, tw2, fc, pd.inv_n); else Rd2::compute(x, tw2, pd, lean);
                if constexpr (NR == 3 && MAC == 2) {
                    Rd2::tensor_epilogue(x, tx, mm, a.tensor_out, period, slot, tile, logn, m, WAVE_PRIVATE ? buf : nullptr, lean, fcp);
                } else if constexpr (NR == 3 && KS) {
                    // the transform of digit k of (o, slot) stays in registers: acc_c += x (.) key[k][c][limb(slot)].  x is lazy, in
                    // [0, 8p); it is only normalised when dl * 8p * p could overflow the 128-bit accumulator (mac_lazy == 0)
                    if constexpr (FP) {
                    } else if (!a.mac_lazy && lean) {
#pragma unroll
                        for (int e = 0; e < 8; e++) x[e] = barrett64(x[e], m);
                    } else if (!a.mac_lazy) {
                        const PrimeConst pc = make_prime_const(pd.p);
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            u64 v[4] = {x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3]};
                            reduce4_from_8p(v, pc);
#pragma unroll
                            for (int i = 0; i < 4; i++) x[4 * h + i] = v[i];
                        }
                    }
                        mac_row(x, kv, mm - m_begin);
                } else if constexpr (NR == 3) {
                    if constexpr (FINAL >= 3) Rd2::template md_write<FINAL>(x, a, mm, slot, tile, logn, m, pd, fcp);
                    else Rd2::template g_write<FINAL>(x, row, tile, logn, m, lean, WAVE_PRIVATE ? buf : nullptr, fcp);
                } else {
                    Rd2::lds_write(x, buf, (FRESH & 16) ? n2_opaque(threadIdx.x) : threadIdx.x);
                    round_sync();
                    if constexpr (!Rd3::HOIST) Rd3::load_tw(tw3, pd, tile, logn, s_first);
                    Rd3::lds_read(x, buf, (FRESH & 32) ? n2_opaque(threadIdx.x) : threadIdx.x);
                    fp_guard(x, 3);
                    if constexpr (FP) Rd3::compute_fp(x, tw3, fc, pd.inv_n); else Rd3::compute(x, tw3, pd, lean);
                    Rd3::template g_write<FINAL>(x, row, tile, logn, m, lean, nullptr, fcp);
                }
            }
        }
        } // !skip_row
        if constexpr (PF) {
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = xn[PF ? e : 0];
        } else if constexpr (!DMA) {
            if (next_wanted) {
                u64 *nrow; const u64 *nin;
                row_ptrs(mm + 1, no, nk, nrow, nin);
                if constexpr (FINAL >= 3) Rd0::template g_read<REDUCE>(x, nin, tile, logn, m, n2_opaque(threadIdx.x)); // offsets formed here, not carried across the epilogue
                else Rd0::template g_read<REDUCE>(x, nin, tile, logn, m);
            }
        }
        ro = no;
        rk = nk;
        if (!skip_row) parity ^= 1u;
    }
    if constexpr (KS) { // one reduction per output coefficient; acc[o][c][slot][N]
        const unsigned o = m_begin / inner;
        const u64 pos = ((u64)tile << N2_LOGT) + 8 * threadIdx.x;
#pragma unroll
        for (int cpt = 0; cpt < 2; cpt++) {
            ulonglong2 *op = reinterpret_cast<ulonglong2 *>(a.mac_acc + ((((u64)o * 2 + cpt) * period + slot) << logn) + pos);
            u64 lin[8];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                ulonglong2 v;
                if constexpr (FP) {
                    v.x = fp_canonical(facc[cpt][2 * e], fc, m.p);
                    v.y = fp_canonical(facc[cpt][2 * e + 1], fc, m.p);
                } else {
                    const Acc128 &p0 = macc[cpt][e >> 1][2 * (e & 1)], &p1 = macc[cpt][e >> 1][2 * (e & 1) + 1];
                    v.x = barrett128(mk64(p0.a0, p0.a1), mk64(p0.a2, p0.a3), m);
                    v.y = barrett128(mk64(p1.a0, p1.a1), mk64(p1.a2, p1.a3), m);
                }
                if constexpr (MAC_STORE_LINEAR) { lin[2 * e] = v.x; lin[2 * e + 1] = v.y; }
                else op[e] = v;
            }
            if constexpr (MAC_STORE_LINEAR) {
                TROY_WAVE_SYNC();
                Rd2::template store_via_lds<(N2_NT & 16) != 0>(lin, a.mac_acc + ((((u64)o * 2 + cpt) * period + slot) << logn), tile, lds[0] + 512 * (threadIdx.x >> 6));
            }
            (void)lin;
        }
    }
}

template <int INV, int STRIDED, int NS, int LOGC, int FINAL, int REDUCE, int MAC = 0>
__global__ __launch_bounds__(STRIDED ? (1 << (NS + LOGC - 3)) : N2_THREADS, MAC == 2 ? N2_TENSOR_WAVES : MAC ? N2_MAC_WAVES : N2_MIN_WAVES) void ntt2_kernel(Ntt2Args a) {
    ntt2_body<INV, STRIDED, NS, LOGC, FINAL, REDUCE, MAC, 0>(a);
}
// the FP64 instances (primes below 2^50): a kernel name of their own, so that profiles and bench.py's per-kernel accounting tell the two apart
template <int INV, int STRIDED, int NS, int LOGC, int FINAL, int REDUCE, int MAC = 0>
__global__ __launch_bounds__(STRIDED ? (1 << (NS + LOGC - 3)) : N2_THREADS, MAC ? N2_FP_MAC_WAVES : N2_MIN_WAVES) void ntt2_fp_kernel(Ntt2Args a) {
    ntt2_body<INV, STRIDED, NS, LOGC, FINAL, REDUCE, MAC, 1>(a);
}

// ---- host side ----
bool ntt2_supported(int logn) { return logn >= 12 && logn <= 17; }

#ifndef TROYHIP_CPU_EMUL
#define N2_KTAG(...) do { if (ktime::enabled) { static thread_local char tagbuf[112]; std::snprintf(tagbuf, sizeof(tagbuf), __VA_ARGS__); ktime::tag = tagbuf; } } while (0)
#else
#define N2_KTAG(...)
#endif
// The forward strided pass is bound by HBM, and the part moves strided rows faster in longer runs: a copy in this pass's pattern reaches 4.9-5.1 TB/s with
// 256-byte runs and 5.7-6.2 TB/s with 512-byte runs (profiles/r05_microbench_nt.txt).  Wide form (N = 2^15): 512 threads own 64 columns x 64 rows
// = 4096 points (64 KiB of LDS with the double buffer, two workgroups per CU); taken when the launch still fills the chip twice over.
#ifndef N2_WIDE
#define N2_WIDE 1
#endif
static bool n2_wide(unsigned narrow_blocks) {
    static const int forced = [] { const char *e = probe_env("TROYHIP_NTT2_WIDE"); return e ? std::atoi(e) : -1; }();
    if (forced >= 0) return forced != 0;
    return N2_WIDE && (narrow_blocks >> 1) >= 4u * device_cus();
}
template <int INV, int STRIDED, int NS, int LOGC, int FINAL, int REDUCE> static void launch_one(const Ntt2Args &a, unsigned blocks, hipStream_t s, bool fp = false) {
    constexpr unsigned THREADS = STRIDED ? (1u << (NS + LOGC - 3)) : N2_THREADS;
    if (fp) {
        N2_KTAG("ntt2_fp_kernel<%d, %d, %d, %d, %d, %d, 0>", INV, STRIDED, NS, LOGC, FINAL, REDUCE);
        TROY_LAUNCH(HIP_KERNEL_NAME(ntt2_fp_kernel<INV, STRIDED, NS, LOGC, FINAL, REDUCE>), dim3(blocks), dim3(THREADS), 0, s, a);
        launch_check("ntt2_fp_kernel");
        return;
    }
    N2_KTAG("ntt2_kernel<%d, %d, %d, %d, %d, %d, 0>", INV, STRIDED, NS, LOGC, FINAL, REDUCE); // the instance's name as rocprofv3 prints it
    TROY_LAUNCH(HIP_KERNEL_NAME(ntt2_kernel<INV, STRIDED, NS, LOGC, FINAL, REDUCE>), dim3(blocks), dim3(THREADS), 0, s, a);
    launch_check("ntt2_kernel");
}
template <int INV, int NS> static void launch_contig(const Ntt2Args &a, unsigned blocks, bool final_pass, hipStream_t s, bool fp = false) {
    if (final_pass) launch_one<INV, 0, NS, 0, INV ? 2 : 1, 0>(a, blocks, s, fp);
    else launch_one<INV, 0, NS, 0, 0, 0>(a, blocks, s, fp);
}
template <int INV, int NS> static void launch_strided(const Ntt2Args &a, unsigned blocks, bool final_pass, bool reduce, hipStream_t s, int md_kind = -1, bool skip_diag = false,
                                                      bool fp = false) {
    constexpr int LOGC = N2_LOGT - NS;
    if (INV) { // the strided pass is the last inverse pass
        if (md_kind == 0) launch_one<1, 1, NS, LOGC, 3, 0>(a, blocks, s, fp);
        else if (md_kind == 2) launch_one<1, 1, NS, LOGC, 4, 0>(a, blocks, s, fp);
        else launch_one<1, 1, NS, LOGC, 2, 0>(a, blocks, s, fp);
    } else {
        (void)final_pass;
        if constexpr (NS == 6) { // N = 2^15: 32 columns = 256-byte runs; the wide form doubles them (n2_wide).  (N = 2^16, NS = 7, 16 -> 32 columns: measured 4 % SLOWER on BGV)
            // not the digit-expanding pass of the key switch (reduce): it writes L + 1 rows per row read and is bound by those writes -- measured
            // 0-4 % SLOWER wide (2105-2110 -> 2118-2150 us at the headline, 2023-2035 -> 2105-2120 on the 49-bit twin); the plain pass: 1609-1636 -> 1432-1438
            if (!reduce && n2_wide(blocks)) {
                Ntt2Args w = a;
                w.tiles_per_row_log = a.tiles_per_row_log - 1;
                stats::counter(stats::NTT2_WIDE_LAUNCHES).fetch_add(1, std::memory_order_relaxed);
                launch_one<0, 1, NS, LOGC + 1, 0, 0>(w, blocks >> 1, s, fp);
                return;
            }
        }
        if (reduce && skip_diag) launch_one<0, 1, NS, LOGC, 0, 2>(a, blocks, s, fp);
        else if (reduce) launch_one<0, 1, NS, LOGC, 0, 1>(a, blocks, s, fp);
        else launch_one<0, 1, NS, LOGC, 0, 0>(a, blocks, s, fp);
    }
}
// stages per round of the two passes, in execution order (forward: Plan<>::r as written; inverse: reversed)
static int pass_rounds(int ns, bool inverse, int (&out)[4]) {
    const int *r = nullptr;
    switch (ns) {
    case 3: r = Plan<3>::r; break;
    case 4: r = Plan<4>::r; break;
    case 5: r = Plan<5>::r; break;
    case 6: r = Plan<6>::r; break;
    case 7: r = Plan<7>::r; break;
    case 9: r = Plan<9>::r; break;
    case 10: r = Plan<10>::r; break;
    default: r = Plan<11>::r; break;
    }
    int n = 0;
    for (int i = 0; i < 4; i++) if (r[i]) n++;
    for (int i = 0; i < n; i++) out[i] = inverse ? r[n - 1 - i] : r[i];
    return n;
}
// the output slots of a launch by class: FP64 instances for the primes below 2^50 (map.fp), the integer instances for the rest
static unsigned select_class(const LimbMap &map, unsigned slot_begin, unsigned slot_count, bool fp, uint8_t (&sel)[64]) {
    unsigned n = 0;
    for (unsigned i = slot_begin; i < slot_begin + slot_count; i++)
        if ((bool)((map.fp >> i) & 1) == fp) sel[n++] = (uint8_t)i;
    return n;
}

// rows are laid out r = (o * period + i) * inner + k; src (optional, forward only): item o, digit k at src + o*src_ostride + k*N
void launch_ntt2(u64 *data, const u64 *src, u64 src_ostride, bool src_reduce, const PrimeDesc *primes, const LimbMap &map, size_t rows, int logn,
                 bool inverse, hipStream_t stream, bool src_same_layout, u64 src_bound) {
    launch_ntt2_slots(data, src, src_ostride, src_reduce, primes, map, rows, logn, inverse, stream, src_same_layout, src_bound, 0, map.period, nullptr);
}
// the general form: only the prime slots [slot_begin, slot_begin + slot_count) of the pattern; md (inverse, inner == 1): the last pass ends in
// the key-switch mod-down instead of storing (Ntt2ModDown, kernels.h).  host_primes (the context's registry, indexed by map.id) switches the
// FP64 instances on for the slots in map.fp: without it every slot takes the integer kernels.
void launch_ntt2_slots(u64 *data, const u64 *src, u64 src_ostride, bool src_reduce, const PrimeDesc *primes, const LimbMap &map, size_t rows, int logn,
                       bool inverse, hipStream_t stream, bool src_same_layout, u64 src_bound, unsigned slot_begin, unsigned slot_count, const Ntt2ModDown *md,
                       unsigned passes, unsigned plan_begin, unsigned plan_count) {
    const u64 *host_primes = map.host_primes;
    if (plan_count && plan_begin + plan_count > map.period) throw Error(ST_INVALID_ARGUMENT, "ntt2: plan slot range");
    if (rows == 0 || slot_count == 0) return;
    if (slot_begin + slot_count > map.period) throw Error(ST_INVALID_ARGUMENT, "ntt2: slot range");
    const bool partial = slot_begin != 0 || slot_count != map.period;
    if ((partial || md || passes != 3) && (!inverse || src)) throw Error(ST_LOGIC_ERROR, "ntt2: slot ranges, single passes and the mod-down epilogue belong to the in-place inverse transform");
    if (md && !(passes & 2)) throw Error(ST_LOGIC_ERROR, "ntt2: the mod-down epilogue rides on the second pass");
    if (md && map.inner != 1) throw Error(ST_LOGIC_ERROR, "ntt2: the mod-down epilogue takes one row per (item, prime)");
    if (!ntt2_supported(logn)) throw Error(ST_LOGIC_ERROR, "ntt2: unsupported size");
    const size_t per_outer = (size_t)map.period * map.inner;
    if (rows % per_outer) throw Error(ST_INVALID_ARGUMENT, "ntt2: row count must be a multiple of the limb pattern");
    if (src && inverse) throw Error(ST_LOGIC_ERROR, "ntt2: out-of-place input is only supported by the forward transform");
    int k2 = 9;
    if (logn - k2 > 7) k2 = logn - 7;
    const int k1 = logn - k2;
    Ntt2Args a;
    a.data = data;
    a.src = nullptr;
    a.src_ostride = 0;
    a.primes = primes;
    a.map = map;
    a.logn = logn;
    a.tiles_per_row_log = (unsigned)(logn - N2_LOGT);
    a.m_total = (unsigned)(rows / per_outer * map.inner);
    // rows per workgroup (they share the prime: twiddles loaded once): up to 8, or the digits of one key-switch group -- fewer when the launch is
    // small, so that a single ciphertext still spreads over the chip (B = 1: 224 workgroups of four rows each left three quarters of the SIMDs'
    // wave slots empty; one row each is 896 workgroups)
    a.rows_per_wg = a.m_total < 8 ? a.m_total : (map.inner > 1 ? (map.inner <= 16 ? map.inner : 8) : 8);
    a.rows_per_wg = plan_per_workgroup(a.m_total, a.rows_per_wg, (size_t)slot_count << a.tiles_per_row_log);
    a.chunks = (a.m_total + a.rows_per_wg - 1) / a.rows_per_wg;
    a.src_reduce = 0;
    a.src_same_layout = 0;
    a.src2 = nullptr; a.src_slots = 0; a.src_split = 0;
    a.slot_fastest = 0;
    a.slot_begin = slot_begin;
    if (md) {
        a.md_ct = md->ct; a.md_ct_bstride = md->ct_bstride; a.md_dl = md->dl; a.md_qk = md->qk; a.md_half = md->half;
        a.md_share = md->share;
        a.md_base = md->base; a.md_base_bstride = md->base_bstride; a.md_base_polys = md->base_polys;
        if (md->kind == 2 && !md->share) throw Error(ST_LOGIC_ERROR, "ntt2: the BGV mod-down needs the special limb's shares");
    }
    // the digit-reducing form of the first pass (unfused key switch) keeps the integer kernels: its FP64 twin lives in launch_ntt2_ks_mac
    const bool fp_ok = host_primes && !(src && src_reduce);
    for (int cls = 0; cls < 2; cls++) {
        const bool fp = cls == 1;
        if (fp && !fp_ok) break;
        a.nsel = 0;
        if (fp_ok) a.nsel = select_class(map, slot_begin, slot_count, fp, a.sel);
        else for (unsigned i = 0; i < slot_count; i++) a.sel[a.nsel++] = (uint8_t)(slot_begin + i);
        if (!a.nsel) continue;
        stats::counter(fp ? stats::NTT2_FP_LAUNCHES : stats::NTT2_INT_LAUNCHES).fetch_add(1, std::memory_order_relaxed);
        unsigned mask1 = 0, mask2 = 0; // reduction sites of the first and of the second pass in execution order (fpmod.h)
        if (fp) {
            u64 pmax = 0;
            for (unsigned i = 0; i < a.nsel; i++) pmax = std::max(pmax, host_primes[map.id[a.sel[i]]]);
            // a pass launched on its own inherits the lazy doubles of a first pass that ran over MORE slots: the walk must assume the largest prime
            // of every slot that shared that pass, or this launch would plan from a smaller bound than the data really carries
            if (plan_count) {
                uint8_t mates[64];
                const unsigned n = select_class(map, plan_begin, plan_count, true, mates);
                for (unsigned i = 0; i < n; i++) pmax = std::max(pmax, host_primes[map.id[mates[i]]]);
            }
            int r1[4], r2[4];
            const int n1 = pass_rounds(inverse ? k2 : k1, inverse, r1), n2 = pass_rounds(inverse ? k1 : k2, inverse, r2);
            const FpPlan p1 = inverse ? fp_plan_inv(pmax, 1.0, r1, n1) : fp_plan(pmax, 1.0, r1, n1);
            const FpPlan p2 = inverse ? fp_plan_inv(pmax, p1.out_bound, r2, n2) : fp_plan(pmax, p1.out_bound, r2, n2);
            if (p1.out_bound < 0 || p2.out_bound < 0) throw Error(ST_LOGIC_ERROR, "ntt2: FP64 bound walk");
            mask1 = p1.mask; mask2 = p2.mask;
        }
        const unsigned blocks = (unsigned)((a.nsel * a.chunks) << a.tiles_per_row_log);
        auto contig = [&](auto inv_tag, const Ntt2Args &args, bool final_pass) {
            constexpr int INV = decltype(inv_tag)::value;
            if (k2 == 9) launch_contig<INV, 9>(args, blocks, final_pass, stream, fp);
            else if (k2 == 10) launch_contig<INV, 10>(args, blocks, final_pass, stream, fp);
            else throw Error(ST_LOGIC_ERROR, "ntt2 plan");
        };
        auto strided = [&](auto inv_tag, const Ntt2Args &args, bool reduce) {
            constexpr int INV = decltype(inv_tag)::value;
            switch (k1) {
            case 3: launch_strided<INV, 3>(args, blocks, false, reduce, stream, md ? md->kind : -1, false, fp); break;
            case 4: launch_strided<INV, 4>(args, blocks, false, reduce, stream, md ? md->kind : -1, false, fp); break;
            case 5: launch_strided<INV, 5>(args, blocks, false, reduce, stream, md ? md->kind : -1, false, fp); break;
            case 6: launch_strided<INV, 6>(args, blocks, false, reduce, stream, md ? md->kind : -1, false, fp); break;
            case 7: launch_strided<INV, 7>(args, blocks, false, reduce, stream, md ? md->kind : -1, false, fp); break;
            default: throw Error(ST_LOGIC_ERROR, "ntt2 plan");
            }
        };
        Ntt2Args first = a, second = a;
        first.fp_red_mask = mask1;
        second.fp_red_mask = mask2;
        if (!inverse) {
            if (src) { first.src = src; first.src_ostride = src_ostride; first.src_reduce = src_reduce; first.src_bound = src_bound; first.src_same_layout = src_same_layout; first.slot_fastest = src_reduce && !src_same_layout; }
            if (passes & 1) strided(std::integral_constant<int, 0>{}, first, src && src_reduce);
            if (passes & 2) contig(std::integral_constant<int, 0>{}, second, true);
        } else {
            if (passes & 1) contig(std::integral_constant<int, 1>{}, first, false);
            // Mod-down epilogue: the rows of DIFFERENT primes read one shared row per item (the special limb; BGV: its 16-byte shares).  Slot-major, the readers
            // of a shared tile were a whole prime's workgroups apart and every one of them fetched it from HBM (traffic 1.51x the rows at N = 2^16).  Slot
            // fastest, they are tiles_per_row workgroups apart -- a multiple of 8, so the hardware's round-robin deals them to the SAME XCD, back to back: one
            // fetch into that L2 serves the class.  (The strided pass's twiddles are a handful of wave-uniform words per prime: nothing to lose in L2.)
            // Same-box A/B at BGV N = 2^16 (profiles/r06_md_order_ab.txt): the FP64 mod-down 3545 -> 2449 us, the workload 4908 -> 5066 ops/s (+3.2 %).
            static const int md_order = [] { const char *e = probe_env("TROYHIP_NTT2_MD_ORDER"); return e ? std::atoi(e) : 1; }();
            if (md) second.slot_fastest = md_order;
            if (passes & 2) strided(std::integral_constant<int, 1>{}, second, false);
        }
    }
}

// Key switching: forward transform of every (digit, output prime) pair with the inner product against the key fused into the
// second pass.  D receives only the first pass; acc [outer][2][period][N] the reduced sums.  Rows are grouped per (o, slot):
// a workgroup takes all `inner` digits of one group, so its 16 accumulators see every term.
// The output primes are split into two classes, each with its own pair of launches: primes below 2^50 (map.fp) run the FP64 instances
// (ntt2_fp_kernel: 8-instruction butterflies, D holds doubles for those slots), the rest the integer instances.  host_primes = the
// context's prime registry (indexed by map.id; digit k is a residue of host_primes[k]).
template <int NS> static void launch_ks_first(const Ntt2Args &first, unsigned blocks, bool fp, bool skip_diag, hipStream_t stream) {
    launch_strided<0, NS>(first, blocks, false, true, stream, -1, skip_diag, fp); // (launch_one names and launches the FP64 instance)
}
void launch_ntt2_ks_mac(u64 *D, const u64 *src, u64 src_ostride, const PrimeDesc *primes, const LimbMap &map, size_t rows, int logn, const u64 *key, u64 *acc,
                        const uint8_t *key_limb, unsigned K, const u64 *ckks_target, u64 t_bstride, bool lazy, u64 src_bound, hipStream_t stream) {
    const u64 *host_primes = map.host_primes;
    if (rows == 0) return;
    if (!ntt2_supported(logn) || logn - 9 > 7 || logn - 9 < 3) throw Error(ST_LOGIC_ERROR, "ntt2 ks_mac: unsupported size");
    const size_t per_outer = (size_t)map.period * map.inner;
    if (rows % per_outer || map.inner > 63) throw Error(ST_INVALID_ARGUMENT, "ntt2 ks_mac: bad row pattern");
    const int k1 = logn - 9;
    Ntt2Args a;
    std::memset(&a, 0, sizeof(a));
    a.data = D;
    a.primes = primes;
    a.map = map;
    a.logn = logn;
    a.tiles_per_row_log = (unsigned)(logn - N2_LOGT);
    a.m_total = (unsigned)(rows / per_outer * map.inner);
    a.rows_per_wg = map.inner;
    a.chunks = a.m_total / a.rows_per_wg;
    const bool skip_diag = ckks_target != nullptr; // CKKS: rows (digit k == output slot) are the NTT-form input, not expanded
    for (int cls = 0; cls < 2; cls++) {
        a.nsel = 0;
        for (unsigned i = 0; i < map.period; i++)
            if ((int)((map.fp >> i) & 1) == cls) a.sel[a.nsel++] = (uint8_t)i;
        if (!a.nsel) continue;
        const bool fp = cls == 1;
        stats::counter(fp ? stats::KS_FP_LAUNCHES : stats::KS_INT_LAUNCHES).fetch_add(1, std::memory_order_relaxed);
        const unsigned blocks = (unsigned)((a.nsel * a.chunks) << a.tiles_per_row_log);
        a.fp_src_wide = 0; a.fp_red_mask = 0; a.fp_acc_every = 0;
        unsigned mask2 = 0;
        if (fp) { // walk the value bound through both passes (fpmod.h): where to reduce, how often the accumulators must be
            if (!host_primes) throw Error(ST_LOGIC_ERROR, "ntt2 ks_mac: the FP64 class needs the host prime list");
            u64 pmax = 0, pmin = ~0ull, src_narrow_max = 0;
            for (unsigned i = 0; i < a.nsel; i++) { const u64 p = host_primes[map.id[a.sel[i]]]; pmax = std::max(pmax, p); pmin = std::min(pmin, p); }
            for (unsigned k = 0; k < map.inner; k++) {
                if (host_primes[k] >> TROY_FP_MAX_BITS) a.fp_src_wide |= u64(1) << k; else src_narrow_max = std::max(src_narrow_max, host_primes[k]);
            }
            const double b_in = std::max(1.0, (double)src_narrow_max / (double)pmin);
            int r1[4], n1 = 0;
            switch (k1) {
            case 3: for (int r : Plan<3>::r) if (r) r1[n1++] = r; break;
            case 4: for (int r : Plan<4>::r) if (r) r1[n1++] = r; break;
            case 5: for (int r : Plan<5>::r) if (r) r1[n1++] = r; break;
            case 6: for (int r : Plan<6>::r) if (r) r1[n1++] = r; break;
            default: for (int r : Plan<7>::r) if (r) r1[n1++] = r; break;
            }
            const FpPlan p1 = fp_plan(pmax, b_in, r1, n1);
            const int r2[3] = {3, 3, 3};
            const FpPlan p2 = fp_plan(pmax, p1.out_bound, r2, 3);
            a.fp_red_mask = p1.mask;
            mask2 = p2.mask;
            const double lim = 0x1p53 / (double)pmax * 0.98, tb = 0.5 + 3.0 * std::max(p2.out_bound, 1.0) * (double)pmax * 0x1p-53;
            const double n = std::floor((lim - 0.5 - 0x1p-40) / tb);
            if (n < 1.0) throw Error(ST_LOGIC_ERROR, "ntt2 ks_mac: FP64 accumulator bound");
            a.fp_acc_every = n >= (double)map.inner ? 0u : (unsigned)n;
        }
        Ntt2Args first = a;
        first.src = src; first.src_ostride = src_ostride; first.src_reduce = 1; first.src_bound = src_bound;
        first.slot_fastest = 1;
        first.skip_diag = skip_diag;
        // the first pass need not keep the digits of a group in one workgroup (only the accumulating second pass does): small launches split them
        first.rows_per_wg = plan_per_workgroup(a.m_total, a.rows_per_wg, (size_t)a.nsel << a.tiles_per_row_log);
        first.chunks = (a.m_total + first.rows_per_wg - 1) / first.rows_per_wg;
        const unsigned blocks1 = (unsigned)((a.nsel * first.chunks) << a.tiles_per_row_log);
        switch (k1) {
        case 3: launch_ks_first<3>(first, blocks1, fp, skip_diag, stream); break;
        case 4: launch_ks_first<4>(first, blocks1, fp, skip_diag, stream); break;
        case 5: launch_ks_first<5>(first, blocks1, fp, skip_diag, stream); break;
        case 6: launch_ks_first<6>(first, blocks1, fp, skip_diag, stream); break;
        default: launch_ks_first<7>(first, blocks1, fp, skip_diag, stream); break;
        }
        Ntt2Args second = a;
        second.fp_red_mask = mask2;
        second.mac_key = key; second.mac_acc = acc; second.mac_target = ckks_target; second.mac_tstride = t_bstride; second.mac_K = K;
        second.mac_lazy = lazy;
        std::memcpy(second.mac_key_limb, key_limb, map.period);
        if (fp) {
            N2_KTAG("ntt2_fp_kernel<0, 0, 9, 0, 1, 0, %d>", skip_diag ? 3 : 1);
            if (skip_diag) TROY_LAUNCH(HIP_KERNEL_NAME(ntt2_fp_kernel<0, 0, 9, 0, 1, 0, 3>), dim3(blocks), dim3(N2_THREADS), 0, stream, second);
            else TROY_LAUNCH(HIP_KERNEL_NAME(ntt2_fp_kernel<0, 0, 9, 0, 1, 0, 1>), dim3(blocks), dim3(N2_THREADS), 0, stream, second);
        } else if (skip_diag) {
            N2_KTAG("ntt2_kernel<0, 0, 9, 0, 1, 0, 3>");
            TROY_LAUNCH(HIP_KERNEL_NAME(ntt2_kernel<0, 0, 9, 0, 1, 0, 3>), dim3(blocks), dim3(N2_THREADS), 0, stream, second);
        } else {
            N2_KTAG("ntt2_kernel<0, 0, 9, 0, 1, 0, 1>");
            TROY_LAUNCH(HIP_KERNEL_NAME(ntt2_kernel<0, 0, 9, 0, 1, 0, 1>), dim3(blocks), dim3(N2_THREADS), 0, stream, second);
        }
        launch_check("ntt2_kernel(ks_mac)");
    }
}

// BFV/BEHZ multiply, both bases: forward transforms of two size-2 operands with the ciphertext tensor fused into the second
// pass.  xa / xb: [batch][2][limbs][N] scratch (transformed in place by the first pass; src_* != nullptr: the first pass reads
// the operand from there instead -- dense ciphertexts are consumed in place); out: [batch][3][limbs][N] in NTT form.
bool ntt2_tensor_supported(int logn) { return ntt2_supported(logn) && logn - 9 >= 3 && logn - 9 <= 7; } // 9-stage second pass
bool ntt2_ks_mac_supported(int logn) { return ntt2_tensor_supported(logn); }
void launch_ntt2_tensor(u64 *xa, const u64 *src_a, u64 *xb, const u64 *src_b, u64 *out, const PrimeDesc *primes, const LimbMap &map, size_t batch, int logn,
                        hipStream_t stream, unsigned src_slots) {
    const u64 *host_primes = map.host_primes;
    if (!batch) return;
    if (!ntt2_tensor_supported(logn) || map.inner != 1) throw Error(ST_LOGIC_ERROR, "ntt2 tensor: unsupported shape");
    const int k1 = logn - 9;
    const bool same = xa == xb; // squaring: one operand, transformed once
    // src_slots != 0: both bases of a BEHZ product in one launch -- the first src_slots slots of every polynomial come from the operands, the rest lie
    // in xa / xb already; when the b-part follows the a-part in memory the first pass of both operands is ONE launch
    const bool joint = src_slots && !same && xb == xa + ((batch * 2 * map.period) << logn);
    if (src_slots && (!src_a || !src_b || src_slots > map.period)) throw Error(ST_LOGIC_ERROR, "ntt2 tensor: operands of the merged form");
    for (int cls = 0; cls < 2; cls++) { // one set of launches per prime class (FP64 instances below 2^50)
        const bool fp = cls == 1;
        if (fp && !host_primes) break;
        uint8_t sel[64];
        unsigned nsel = 0;
        if (host_primes) nsel = select_class(map, 0, map.period, fp, sel);
        else for (unsigned i = 0; i < map.period; i++) sel[nsel++] = (uint8_t)i;
        if (!nsel) continue;
        stats::counter(fp ? stats::NTT2_FP_LAUNCHES : stats::NTT2_INT_LAUNCHES).fetch_add(1, std::memory_order_relaxed);
        unsigned mask1 = 0, mask2 = 0;
        if (fp) {
            u64 pmax = 0;
            for (unsigned i = 0; i < nsel; i++) pmax = std::max(pmax, host_primes[map.id[sel[i]]]);
            int r1[4], r2[4];
            const int n1 = pass_rounds(k1, false, r1), n2 = pass_rounds(9, false, r2);
            const FpPlan p1 = fp_plan(pmax, 1.0, r1, n1), p2 = fp_plan(pmax, p1.out_bound, r2, n2);
            mask1 = p1.mask; mask2 = p2.mask;
        }
        for (int part = 0; part < (same || joint ? 1 : 2); part++) { // first pass of both operands (rows = batch * 2 * limbs each)
            Ntt2Args a;
            std::memset(&a, 0, sizeof(a));
            a.data = part ? xb : xa;
            const u64 *src = part ? src_b : src_a;
            a.primes = primes;
            a.map = map;
            a.logn = logn;
            a.tiles_per_row_log = (unsigned)(logn - N2_LOGT);
            a.m_total = (unsigned)(batch * (joint ? 4 : 2));
            a.nsel = nsel;
            std::memcpy(a.sel, sel, sizeof(sel));
            a.fp_red_mask = mask1;
            a.rows_per_wg = plan_per_workgroup(a.m_total, a.m_total < 8 ? a.m_total : 8, (size_t)nsel << a.tiles_per_row_log);
            a.chunks = (a.m_total + a.rows_per_wg - 1) / a.rows_per_wg;
            if (src_slots) { a.src = src; a.src2 = src_b; a.src_same_layout = 2; a.src_slots = src_slots; a.src_split = joint ? (unsigned)(batch * 2) : ~0u; }
            else if (src) { a.src = src; a.src_same_layout = 1; }
            const unsigned blocks = (unsigned)((nsel * a.chunks) << a.tiles_per_row_log);
            switch (k1) {
            case 3: launch_strided<0, 3>(a, blocks, false, false, stream, -1, false, fp); break;
            case 4: launch_strided<0, 4>(a, blocks, false, false, stream, -1, false, fp); break;
            case 5: launch_strided<0, 5>(a, blocks, false, false, stream, -1, false, fp); break;
            case 6: launch_strided<0, 6>(a, blocks, false, false, stream, -1, false, fp); break;
            default: launch_strided<0, 7>(a, blocks, false, false, stream, -1, false, fp); break;
            }
        }
        Ntt2Args a;
        std::memset(&a, 0, sizeof(a));
        a.data = xa; a.tensor_b = xb; a.tensor_out = out;
        a.primes = primes;
        a.map = map;
        a.logn = logn;
        a.tiles_per_row_log = (unsigned)(logn - N2_LOGT);
        a.m_total = (unsigned)(batch * 4);
        a.nsel = nsel;
        std::memcpy(a.sel, sel, sizeof(sel));
        a.fp_red_mask = mask2;
        a.rows_per_wg = 4;
        a.chunks = (unsigned)batch;
        const unsigned blocks = (unsigned)((nsel * a.chunks) << a.tiles_per_row_log);
        if (fp) {
            N2_KTAG("ntt2_fp_kernel<0, 0, 9, 0, 1, 0, 2>");
            TROY_LAUNCH(HIP_KERNEL_NAME(ntt2_fp_kernel<0, 0, 9, 0, 1, 0, 2>), dim3(blocks), dim3(N2_THREADS), 0, stream, a);
        } else {
            N2_KTAG("ntt2_kernel<0, 0, 9, 0, 1, 0, 2>");
            TROY_LAUNCH(HIP_KERNEL_NAME(ntt2_kernel<0, 0, 9, 0, 1, 0, 2>), dim3(blocks), dim3(N2_THREADS), 0, stream, a);
        }
        launch_check("ntt2_kernel(tensor)");
    }
}

} // namespace troyhip
