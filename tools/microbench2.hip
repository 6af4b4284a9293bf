// microbench2.hip -- round-2 VALU probes: butterfly formulations (guarded vs guard-free, mul_acc forms) at 8 and 4 waves
// per SIMD, plus the issue cost of the helper instructions the transforms use.
// Build: hipcc --offload-arch=gfx950 -O3 -I troy_amd/csrc tools/microbench2.hip -o tools/microbench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../troy_amd/csrc/bfly.h"

using namespace troyhip;
#define ITERS 2048

// acc + w*y + q*negp (mod 2^64) with the four cross products chained through one v_mad_u64_u32 accumulator (only its low
// word is used) instead of 4 v_mul_lo_u32 + 2 v_add3_u32
__device__ __forceinline__ u64 mul_acc_chain(u64 acc, u64 y, u64 w, u64 q, u64 negp) {
    const u32 y0 = lo32(y), y1 = hi32(y), w0 = lo32(w), w1 = hi32(w), d0 = lo32(q), d1 = hi32(q);
    u64 t, r, sc;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(t), "=s"(sc) : "v"(w0), "v"(y1));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(t), "=s"(sc) : "v"(w1), "v"(y0));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(t), "=s"(sc) : "v"(d0), "s"(hi32(negp)));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(t), "=s"(sc) : "v"(d1), "s"(lo32(negp)));
    const u64 a = mk64(lo32(acc), hi32(acc) + lo32(t));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(sc) : "v"(w0), "v"(y0), "v"(a));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(r), "=s"(sc) : "v"(d0), "s"(lo32(negp)));
    return r;
}
__device__ __forceinline__ void ct_bfly4_chain(u64 (&X)[4], u64 (&Y)[4], const Shoup (&w)[4], const PrimeConst &c) {
    u64 q[4], xn[4], t[4];
    mulhi_approx4(q, Y, w);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        xn[i] = mul_acc_chain(X[i], Y[i], w[i].op, q[i], c.negp);
        t[i] = (X[i] << 1) + c.three_p;
    }
    sub4(Y, t, xn);
#pragma unroll
    for (int i = 0; i < 4; i++) X[i] = xn[i];
}

// one radix-8 round (12 butterflies on 8 values) per iteration
template <int VAR> __global__ __launch_bounds__(256) void bfly_probe(u64 *out, u64 p, Shoup w) {
    extern __shared__ u64 dyn[];
    const u32 t = threadIdx.x + blockIdx.x * 256;
    u64 x[8];
    const PrimeConst pc = make_prime_const(p);
    for (int i = 0; i < 8; i++) x[i] = (t * 0x9E3779B97F4A7C15ull + i * 0xBF58476D1CE4E5B9ull) % p;
    Shoup ww[4] = {w, w, w, w};
    const u64 kp = 8 * p;
    for (int it = 0; it < ITERS / 16; it++) {
#pragma unroll
        for (int st = 0; st < 3; st++) {
            const int half = 4 >> st;
            u64 X[4], Y[4]; int n = 0, ix[4], iy[4];
#pragma unroll
            for (int blk = 0; blk < (1 << st); blk++)
#pragma unroll
                for (int k = 0; k < half; k++) { ix[n] = blk * 2 * half + k; iy[n] = ix[n] + half; n++; }
#pragma unroll
            for (int i = 0; i < 4; i++) { X[i] = x[ix[i]]; Y[i] = x[iy[i]]; }
            if (VAR == 0) ct_bfly4(X, Y, ww, pc);
            if (VAR == 1) ct_bfly4_ng(X, Y, ww, pc);
            if (VAR == 2) gs_bfly4(X, Y, ww, pc);
            if (VAR == 3) gs_bfly4_ng(X, Y, ww, kp, pc);
            if (VAR == 4) ct_bfly4_chain(X, Y, ww, pc);
#pragma unroll
            for (int i = 0; i < 4; i++) { x[ix[i]] = X[i]; x[iy[i]] = Y[i]; }
        }
    }
    u64 r = 0;
    for (int i = 0; i < 8; i++) r ^= x[i];
    if (r == 0x1234567) dyn[threadIdx.x] = r;
    out[t] = r;
}

#define CH 8
template <int OP> __global__ __launch_bounds__(256) void probe(u32 *out, u32 sb) {
    u32 t = threadIdx.x + blockIdx.x * 256;
    u32 a[CH];
    u64 w[CH];
    u32 b = t * 2654435761u + 12345u, c = t ^ 0x9E3779B9u;
    u64 m = 0x5555555555555555ull ^ sb;
    for (int i = 0; i < CH; i++) { a[i] = t + i * 77; w[i] = ((u64)t << 32) | (i * 1315423911u); }
    for (int it = 0; it < ITERS * 2; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m));
            if (OP == 2) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 3) asm volatile("v_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 4) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
            if (OP == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 6) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 7) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(b), "s"(sb) : "vcc");
            if (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(sb));
            if (OP == 9) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w[i]) : "v"(w[(i + 1) % CH]));
            if (OP == 10) asm volatile("v_add_co_u32 %0, %1, %0, %2" : "+v"(a[i]), "=s"(m) : "v"(b));
            if (OP == 11) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "vcc");
            if (OP == 12) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 13) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 14) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) % CH]));
            if (OP == 15) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 16) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(w[i]));
            if (OP == 17) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    u32 r = (u32)m;
    for (int i = 0; i < CH; i++) r ^= a[i] ^ (u32)w[i] ^ (u32)(w[i] >> 32);
    out[t] = r;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
    u32 *out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto timeit = [&](auto &&launch) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return best;
    };
    const char *names[] = {"v_add_u32", "v_cndmask_b32(sgpr)", "v_sub_co_u32", "v_subb_co_u32", "v_lshlrev_b32", "v_xor_b32", "v_mov_b32", "v_mad_u64_u32(v,s)",
                           "v_mul_lo_u32(v,s)", "v_lshl_add_u64", "v_add_co_u32(sgpr)", "v_mad_u64_u32", "v_add3_u32", "v_mul_hi_u32", "v_permlane32_swap",
                           "v_and_or_b32", "v_lshrrev_b64", "v_pk_add_u16"};
    const int blocks = cus * 8;
    double base = 0;
    for (int op = 0; op < 18; op++) {
        float ms = timeit([&] {
            switch (op) {
#define C(n) case n: probe<n><<<blocks, 256>>>(out, 12345u); break;
                C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17)
#undef C
            }
        });
        double wave_instr = (double)blocks * 4 * ITERS * 2 * CH;
        double cyc = clk / (wave_instr / (cus * 4.0) / (ms * 1e-3));
        if (op == 0) base = cyc;
        printf("%-22s %8.3f ms -> %6.2f nominal cycles/wave-instr/SIMD (x%.2f of v_add_u32)\n", names[op], ms, cyc, cyc / base);
    }
    const u64 p = 288230376150630401ull; // a 58-bit prime = 1 mod 2^16
    const u64 wv = 0x123456789abcdefULL % p;
    const Shoup sw{wv, (u64)((((unsigned __int128)wv) << 64) / p)};
    const char *vn[] = {"ct_bfly4 (guarded)", "ct_bfly4_ng", "gs_bfly4 (guarded)", "gs_bfly4_ng", "ct_bfly4_ng mad-chain"};
    for (int occ = 0; occ < 2; occ++) {
        const size_t lds = occ ? 40 * 1024 : 0; // 40 KiB per 256-thread block -> 4 blocks per CU -> 4 waves per SIMD
        for (int v = 0; v < 5; v++) {
            float ms = timeit([&] {
                switch (v) {
#define C(n) case n: bfly_probe<n><<<blocks, 256, lds>>>((u64 *)out, p, sw); break;
                    C(0) C(1) C(2) C(3) C(4)
#undef C
                }
            });
            double bf = (double)blocks * 256 * (ITERS / 16) * 12;
            double per_s = bf / (ms * 1e-3);
            printf("%-24s %s waves/SIMD %8.3f ms -> %7.1f G butterflies/s chip-wide, %.2f nominal cycles per wave-butterfly per SIMD; N=2^15 limb floor %.4f us\n", vn[v],
                   occ ? "4" : "8", ms, per_s / 1e9, clk / (per_s / 64 / (cus * 4.0)), 245760.0 / per_s * 1e6);
        }
    }
    return 0;
}
