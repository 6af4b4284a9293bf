// ntt.hip -- batched negacyclic NTT / INTT for gfx950 (the roofline kernel of SURVEY.md section 8a-2).
//
// Replaces the reference's one-launch-per-stage radix-2 kernels gNttTransferToRevLayered /
// gNttTransferFromRevLayered + gMultiplyInvDegreeNttTables + gModBoundedUsingNttTables
// (src/kernelutils.cu:330-493, 256-270, 201-216).  Same mathematical transform as the CPU path
// (src/utils/dwthandler.h:88-372, tables src/utils/ntt.cpp:17-66): forward = Cooley-Tukey,
// natural -> bit-reversed order; inverse = Gentleman-Sande, bit-reversed -> natural, N^-1 folded into
// the last stage.  Outputs are stored as canonical residues in [0,p).
//
// MI355X design: a transform is split into at most two passes.  Each pass is ONE launch over the
// whole batch (every limb of every polynomial of every ciphertext): a 256-thread workgroup stages a
// 2048-coefficient tile (16 KiB) in LDS with 16-byte coalesced loads, runs up to 11 butterfly stages
// on it as radix-8 register rounds (3 stages per LDS round trip), and writes it back coalesced.
//   pass "strided":    the first k1 stages act on points N/2^k1 apart; a tile is 2^k1 points x C
//                      contiguous columns (C*8 >= 128 bytes per segment).
//   pass "contiguous": the remaining stages act inside contiguous blocks of 2^k2 coefficients.
// Algorithmic traffic: 16 B per coefficient per transform; actual: 16 B per pass.
#include "kernels.h"

namespace troyhip {

struct NttArgs {
    u64 *data;
    const PrimeDesc *primes;
    LimbMap map;
    int logn;
    int logT;       // log2 of tile size (min(logn, 11))
    int s_first;    // first global stage of this pass (forward: lowest s; inverse: highest s)
    int n_stages;
    int logC;       // log2 columns (strided pass) or 0
    int lps;        // log2 point stride of the strided pass, 0 for contiguous
    int strided;
    int tiles_per_row_log;
    int final_pass; // apply final range reduction
};

#define NTT_THREADS 256

__device__ __forceinline__ u64 gidx(const NttArgs &a, unsigned tile, unsigned f) {
    unsigned base = a.strided ? (tile << a.logC) : (tile << a.logT);
    return ((u64)(f >> a.logC) << a.lps) + base + (f & ((1u << a.logC) - 1));
}

__device__ __forceinline__ void ct_bfly(u64 &X, u64 &Y, const Shoup w, u64 p, u64 two_p) {
    u64 u = X >= two_p ? X - two_p : X;
    u64 v = mul_lazy(Y, w.op, w.quo, p);
    X = u + v;
    Y = u + two_p - v;
}
__device__ __forceinline__ void gs_bfly(u64 &X, u64 &Y, const Shoup w, u64 p, u64 two_p) {
    u64 u = X, v = Y;
    u64 s = u + v;
    X = s >= two_p ? s - two_p : s;
    Y = mul_lazy(u + two_p - v, w.op, w.quo, p);
}
// last inverse stage with N^-1 folded in (src/utils/dwthandler.h:289-330)
__device__ __forceinline__ void gs_bfly_last(u64 &X, u64 &Y, const Shoup w_scaled, const Shoup inv_n, u64 p, u64 two_p) {
    u64 u = X, v = Y;
    u64 s = u + v;
    s = s >= two_p ? s - two_p : s;
    X = mul_lazy(s, inv_n.op, inv_n.quo, p);
    Y = mul_lazy(u + two_p - v, w_scaled.op, w_scaled.quo, p);
}

// R forward stages on 2^R register-resident points; idx1 = twiddle index of the first stage
template <int R> __device__ __forceinline__ void fwd_round(u64 *x, const Shoup *root, unsigned idx1, u64 p, u64 two_p) {
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int half = (1 << R) >> (st + 1);
#pragma unroll
        for (int blk = 0; blk < (1 << st); blk++) {
            const Shoup w = root[(idx1 << st) + blk];
#pragma unroll
            for (int k = 0; k < half; k++) ct_bfly(x[blk * 2 * half + k], x[blk * 2 * half + k + half], w, p, two_p);
        }
    }
}
// R inverse stages (global stages s, s-1, ..); blk0 = block index of x[0] at stage s
template <int R> __device__ __forceinline__ void inv_round(u64 *x, const PrimeDesc &pd, int logn, int s, unsigned blk0, u64 p, u64 two_p) {
    const unsigned n = 1u << logn;
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int dist = 1 << st;
        const int cs = s - st; // current global stage, m = 2^cs groups
        const unsigned tbase = n - (2u << cs) + 1 + (blk0 >> st);
#pragma unroll
        for (int blk = 0; blk < ((1 << R) >> (st + 1)); blk++) {
            if (cs == 0) {
#pragma unroll
                for (int k = 0; k < dist; k++) gs_bfly_last(x[blk * 2 * dist + k], x[blk * 2 * dist + k + dist], pd.iroot_last_scaled, pd.inv_n, p, two_p);
            } else {
                const Shoup w = pd.iroot[tbase + blk];
#pragma unroll
                for (int k = 0; k < dist; k++) gs_bfly(x[blk * 2 * dist + k], x[blk * 2 * dist + k + dist], w, p, two_p);
            }
        }
    }
}

template <int R> __device__ __forceinline__ void fwd_round_lds(u64 *lds, const NttArgs &a, const PrimeDesc &pd, unsigned tile, int local_stage, unsigned T) {
    const int s = a.s_first + local_stage;
    const unsigned logPf = a.n_stages + a.logC;
    const unsigned logg = logPf - local_stage - 1; // log2 of the first stage's gap (flattened)
    const unsigned logd = logg - (R - 1);          // log2 distance between a thread's points
    for (unsigned q = threadIdx.x; q < (T >> R); q += NTT_THREADS) {
        unsigned hi = q >> logd, lo = q & ((1u << logd) - 1);
        unsigned base = (hi << (logg + 1)) + lo;
        u64 x[1 << R];
#pragma unroll
        for (int e = 0; e < (1 << R); e++) x[e] = lds[base + (e << logd)];
        unsigned idx1 = (1u << s) + (unsigned)(gidx(a, tile, base) >> (a.logn - s));
        fwd_round<R>(x, pd.root, idx1, pd.p, pd.two_p);
#pragma unroll
        for (int e = 0; e < (1 << R); e++) lds[base + (e << logd)] = x[e];
    }
}
template <int R> __device__ __forceinline__ void inv_round_lds(u64 *lds, const NttArgs &a, const PrimeDesc &pd, unsigned tile, int local_stage, unsigned T) {
    const int s = a.s_first - local_stage; // global stage of the first (smallest-gap) stage in the round
    const unsigned logd = local_stage + a.logC; // flattened gap of that stage
    for (unsigned q = threadIdx.x; q < (T >> R); q += NTT_THREADS) {
        unsigned hi = q >> logd, lo = q & ((1u << logd) - 1);
        unsigned base = (hi << (logd + R)) + lo;
        u64 x[1 << R];
#pragma unroll
        for (int e = 0; e < (1 << R); e++) x[e] = lds[base + (e << logd)];
        unsigned blk0 = (unsigned)(gidx(a, tile, base) >> (a.logn - s));
        inv_round<R>(x, pd, a.logn, s, blk0, pd.p, pd.two_p);
#pragma unroll
        for (int e = 0; e < (1 << R); e++) lds[base + (e << logd)] = x[e];
    }
}

template <int INVERSE> __global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(NttArgs a) {
    __shared__ u64 lds[2048];
    const unsigned T = 1u << a.logT;
    const unsigned tile = blockIdx.x & ((1u << a.tiles_per_row_log) - 1);
    const unsigned row = blockIdx.x >> a.tiles_per_row_log;
    const PrimeDesc pd = a.primes[a.map.id[(row / a.map.inner) % a.map.period]];
    u64 *x = a.data + ((u64)row << a.logn);

    if (T >= 2) {
        for (unsigned f = threadIdx.x * 2; f < T; f += 2 * NTT_THREADS) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(x + gidx(a, tile, f));
            lds[f] = v.x;
            lds[f + 1] = v.y;
        }
    }
    __syncthreads();

    int st = 0;
    for (; st + 3 <= a.n_stages; st += 3) {
        if (INVERSE) inv_round_lds<3>(lds, a, pd, tile, st, T); else fwd_round_lds<3>(lds, a, pd, tile, st, T);
        __syncthreads();
    }
    if (a.n_stages - st == 2) {
        if (INVERSE) inv_round_lds<2>(lds, a, pd, tile, st, T); else fwd_round_lds<2>(lds, a, pd, tile, st, T);
        __syncthreads();
    } else if (a.n_stages - st == 1) {
        if (INVERSE) inv_round_lds<1>(lds, a, pd, tile, st, T); else fwd_round_lds<1>(lds, a, pd, tile, st, T);
        __syncthreads();
    }

    const u64 p = pd.p, two_p = pd.two_p;
    for (unsigned f = threadIdx.x * 2; f < T; f += 2 * NTT_THREADS) {
        ulonglong2 v;
        v.x = lds[f];
        v.y = lds[f + 1];
        if (a.final_pass) {
            if (!INVERSE) {
                v.x = v.x >= two_p ? v.x - two_p : v.x;
                v.y = v.y >= two_p ? v.y - two_p : v.y;
            }
            v.x = v.x >= p ? v.x - p : v.x;
            v.y = v.y >= p ? v.y - p : v.y;
        }
        *reinterpret_cast<ulonglong2 *>(x + gidx(a, tile, f)) = v;
    }
}

// ---- host side ----
// Stage split: the contiguous pass takes k2 stages, the strided pass k1 = logn - k2 (0 when N <= 2048).
static void plan(int logn, int &k1, int &k2) {
    if (logn <= 11) { k1 = 0; k2 = logn; return; }
    k2 = 9;                       // 3 radix-8 rounds
    if (logn - k2 > 7) k2 = logn - 7; // keep >= 16 columns (128-byte segments) in the strided pass
    k1 = logn - k2;
}

void launch_ntt_from(u64 *data, const u64 *src, const PrimeDesc *primes, const LimbMap &map, size_t rows, int logn, hipStream_t stream) {
    if (rows == 0) return;
    if (ntt1_supported(logn, map, rows)) { launch_ntt1(data, src, primes, map, rows, logn, false, stream); return; }
    if (ntt2_supported(logn) && rows % ((size_t)map.period * map.inner) == 0) { // the first pass reads src, no copy
        launch_ntt2(data, src, 0, false, primes, map, rows, logn, false, stream, true);
        return;
    }
    HIP_CHECK(hipMemcpyAsync(data, src, (rows << logn) * sizeof(u64), hipMemcpyDeviceToDevice, stream));
    launch_ntt(data, primes, map, rows, logn, false, stream);
}
void launch_ntt(u64 *data, const PrimeDesc *primes, const LimbMap &map, size_t rows, int logn, bool inverse, hipStream_t stream) {
    if (rows == 0) return;
    if (ntt1_supported(logn, map, rows)) { launch_ntt1(data, nullptr, primes, map, rows, logn, inverse, stream); return; } // N = 2^12 .. 2^15: single pass
    if (ntt2_supported(logn) && rows % ((size_t)map.period * map.inner) == 0) { // production path for N >= 4096
        launch_ntt2(data, nullptr, 0, false, primes, map, rows, logn, inverse, stream);
        return;
    }
    int k1, k2;
    plan(logn, k1, k2);
    NttArgs a;
    a.data = data;
    a.primes = primes;
    a.map = map;
    a.logn = logn;
    a.logT = logn < 11 ? logn : 11;
    a.tiles_per_row_log = logn - a.logT;
    unsigned blocks = (unsigned)(rows << a.tiles_per_row_log);
    auto strided = [&](bool final_pass) {
        a.strided = 1; a.n_stages = k1; a.logC = a.logT - k1; a.lps = logn - k1; a.final_pass = final_pass;
        a.s_first = inverse ? k1 - 1 : 0;
    };
    auto contiguous = [&](bool final_pass) {
        a.strided = 0; a.n_stages = k2; a.logC = 0; a.lps = 0; a.final_pass = final_pass;
        a.s_first = inverse ? logn - 1 : k1;
    };
    if (!inverse) {
        if (k1) { strided(false); TROY_LAUNCH(HIP_KERNEL_NAME(ntt_pass_kernel<0>), dim3(blocks), dim3(NTT_THREADS), 0, stream, a); launch_check("ntt_fwd_strided"); }
        contiguous(true);
        TROY_LAUNCH(HIP_KERNEL_NAME(ntt_pass_kernel<0>), dim3(blocks), dim3(NTT_THREADS), 0, stream, a);
        launch_check("ntt_fwd_contiguous");
    } else {
        contiguous(k1 == 0);
        TROY_LAUNCH(HIP_KERNEL_NAME(ntt_pass_kernel<1>), dim3(blocks), dim3(NTT_THREADS), 0, stream, a);
        launch_check("ntt_inv_contiguous");
        if (k1) { strided(true); TROY_LAUNCH(HIP_KERNEL_NAME(ntt_pass_kernel<1>), dim3(blocks), dim3(NTT_THREADS), 0, stream, a); launch_check("ntt_inv_strided"); }
    }
}

} // namespace troyhip
