// microbench_nt.hip -- what the cache policy of streaming accesses is worth on gfx950: a copy / scale kernel over 4 GiB (16 bytes per lane, grid-stride in
// 256 KiB rows like the limb kernels) with plain or non-temporal loads and stores, and with a small table every workgroup re-reads (does the stream evict it?).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_nt.hip -o tools/microbench_nt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef u64 v2 __attribute__((ext_vector_type(2)));

template <int POL> __global__ __launch_bounds__(256) void copy_kernel(const v2 *in, v2 *out, const v2 *table, size_t rows, int use_table, int rotate = 0) {
    // one workgroup per 256 KiB row at a time (16384 x 16 B), 64 iterations of 256 threads; rotate: every workgroup starts its rows at another 4 KiB piece
    const unsigned rot = rotate ? (blockIdx.x * 7u) & 63u : 0u;
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const v2 *src = in + r * 16384;
        v2 *dst = out + r * 16384;
#pragma unroll 4
        for (int i = 0; i < 64; i++) {
            const unsigned k = threadIdx.x + 256 * ((i + rot) & 63u);
            v2 v = (POL & 1) ? __builtin_nontemporal_load(src + k) : src[k];
            if (use_table) { const v2 t = table[k & 32767]; v.x ^= t.x; v.y += t.y; } // a 512 KiB table shared by every row
            if (POL & 2) __builtin_nontemporal_store(v, dst + k); else dst[k] = v;
        }
    }
}
// the access pattern of the strided first pass (ntt2.hip, N = 2^15): a workgroup owns 32 columns x 64 rows of a limb = 64 runs of 256 bytes at a stride of 4 KiB,
// 8-byte accesses, 8 per thread (thread t: column t % 32, rows t / 32 + 8 e); SEG = 64: the same with 64 columns x 32 rows (512-byte runs)
template <int POL, int COLS> __global__ __launch_bounds__(256) void strided_kernel(const u64 *in, u64 *out, size_t tiles) {
    constexpr int ROWS = 2048 / COLS, TPL = 32768 / 2048; // tiles per limb
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t limb = t / TPL, tile = t % TPL;
        const u64 *src = in + limb * 32768 + tile * COLS;
        u64 *dst = out + limb * 32768 + tile * COLS;
        u64 v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const unsigned idx = (threadIdx.x / COLS + (256 / COLS) * e) * (32768 / ROWS) + threadIdx.x % COLS;
            v[e] = (POL & 1) ? __builtin_nontemporal_load(src + idx) : src[idx];
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const unsigned idx = (threadIdx.x / COLS + (256 / COLS) * e) * (32768 / ROWS) + threadIdx.x % COLS;
            if (POL & 2) __builtin_nontemporal_store(v[e] + 1, dst + idx); else dst[idx] = v[e] + 1;
        }
    }
}
// the access pattern of the BEHZ conversions: a workgroup walks tiles of RUN bytes of ONE polynomial, reading the tile's position in 14 limb rows (256 KiB apart)
// and writing it in 15 -- 29 streams with a stride of a limb.  RUN = 512 is the library's (64 coefficients); longer runs = fewer DRAM page visits per byte.
template <int RUN> __global__ __launch_bounds__(256) void limbs_kernel(const u64 *in, u64 *out, size_t polys) {
    constexpr int W = RUN / 8;                 // words per run
    constexpr int TILES = 32768 / W;           // tiles per polynomial
    for (size_t t = blockIdx.x; t < polys * TILES; t += gridDim.x) {
        const size_t poly = t / TILES, tile = t % TILES;
        const u64 *src = in + poly * 14 * 32768 + tile * W;
        u64 *dst = out + poly * 15 * 32768 + tile * W;
        // 256 threads: thread t handles word t % W of limbs t / W, t / W + 256 / W, ...
        u64 acc = 0;
        for (int l = threadIdx.x / W; l < 14; l += 256 / W) acc += src[(size_t)l * 32768 + threadIdx.x % W];
        for (int l = threadIdx.x / W; l < 15; l += 256 / W) dst[(size_t)l * 32768 + threadIdx.x % W] = acc + l;
    }
}
int main() {
    const size_t bytes = (size_t)4 << 30, rows = bytes / (256 << 10);
    v2 *in, *out, *table;
    hipMalloc(&in, bytes); hipMalloc(&out, bytes); hipMalloc(&table, 512 << 10);
    hipMemset(in, 1, bytes); hipMemset(table, 3, 512 << 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[4] = {"plain loads, plain stores", "NT loads,    plain stores", "plain loads, NT stores", "NT loads,    NT stores"};
    for (int grid : {256 * 8, 256 * 4, 256 * 2})
        for (int tab = 0; tab < 2; tab++)
            for (int pol = 0; pol < 4; pol++) {
                float best = 1e30f;
                for (int rep = 0; rep < 5; rep++) {
                    hipEventRecord(e0);
                    switch (pol) {
                    case 0: copy_kernel<0><<<grid, 256>>>(in, out, table, rows, tab); break;
                    case 1: copy_kernel<1><<<grid, 256>>>(in, out, table, rows, tab); break;
                    case 2: copy_kernel<2><<<grid, 256>>>(in, out, table, rows, tab); break;
                    case 3: copy_kernel<3><<<grid, 256>>>(in, out, table, rows, tab); break;
                    }
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("grid %5d  %s  %s  %7.3f ms  %6.2f TB/s (read + write)\n", grid, tab ? "with table" : "no table  ", names[pol], best, 2.0 * bytes / best / 1e9);
            }
    {
        const size_t polys = bytes / (15 * 32768 * 8);
        for (int grid : {256 * 16, 256 * 8})
            for (int run : {512, 1024, 2048}) {
                float best = 1e30f;
                for (int rep = 0; rep < 5; rep++) {
                    hipEventRecord(e0);
                    if (run == 512) limbs_kernel<512><<<grid, 256>>>((const u64 *)in, (u64 *)out, polys);
                    else if (run == 1024) limbs_kernel<1024><<<grid, 256>>>((const u64 *)in, (u64 *)out, polys);
                    else limbs_kernel<2048><<<grid, 256>>>((const u64 *)in, (u64 *)out, polys);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("limb streams (14 in, 15 out, 256 KiB apart)  runs of %4d bytes  grid %5d  %7.3f ms  %6.2f TB/s (read + write)\n", run, grid, best, polys * 29.0 * 32768 * 8 / best / 1e9);
            }
    }
    for (int rot = 0; rot < 2; rot++)
        for (int grid : {256 * 8, 256 * 4}) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(e0);
                copy_kernel<2><<<grid, 256>>>(in, out, table, rows, 0, rot);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("contiguous rows, NT stores, %s  grid %5d  %7.3f ms  %6.2f TB/s (read + write)\n", rot ? "rotated start per workgroup" : "every workgroup from offset 0 ", grid, best, 2.0 * bytes / best / 1e9);
        }
    for (int cols : {32, 64})
        for (int grid : {256 * 16, 256 * 8, 256 * 4})
            for (int pol = 0; pol < 4; pol++) {
                float best = 1e30f;
                const size_t tiles = bytes / (2048 * 8);
                for (int rep = 0; rep < 5; rep++) {
                    hipEventRecord(e0);
#define L(P, C) strided_kernel<P, C><<<grid, 256>>>((const u64 *)in, (u64 *)out, tiles)
                    if (cols == 32) { switch (pol) { case 0: L(0, 32); break; case 1: L(1, 32); break; case 2: L(2, 32); break; case 3: L(3, 32); break; } }
                    else { switch (pol) { case 0: L(0, 64); break; case 1: L(1, 64); break; case 2: L(2, 64); break; case 3: L(3, 64); break; } }
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("strided %3d-byte runs  grid %5d  %s  %7.3f ms  %6.2f TB/s (read + write)\n", cols * 8, grid, names[pol], best, 2.0 * bytes / best / 1e9);
            }
    return 0;
}
