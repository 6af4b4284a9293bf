// mfma_overlap.hip -- do int8 MFMAs and 64-bit integer VALU work overlap on gfx950, and under which issue pattern?
// (development tool for behz.hip).  Each wave runs ITER rounds of {4 chained v_mfma_i32_32x32x32_i8, NV v_mad_u64_u32};
//   mode 0: MFMA only   mode 1: VALU only   mode 2: MFMA then the VALU block that consumes its result (serial, as behz.hip)
//   mode 4: mode 3 with sched_group_barrier interleaving MFMA and VALU instruction by instruction
//   mode 3: software-pipelined -- the MFMAs of round i+1 are issued before the VALU block of round i (two accumulators)
// build + run: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_overlap.hip -o gpurun_out/mfma_overlap && gpurun_out/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;
template <int NV> __device__ __forceinline__ u64 valu_block(u64 x, const v16i &c, unsigned k) {
#pragma unroll
#ifdef ADDONLY // -DADDONLY: the VALU block is shifts / adds / xors only (no multiplier)
    for (int i = 0; i < NV; i++) x = ((x << 3) ^ (u64)(unsigned)c[i & 15]) + (x >> 29) + k;
#else
    for (int i = 0; i < NV; i++) x = (u64)(unsigned)x * k + (u64)(unsigned)c[i & 15] + (x >> 32);
#endif
    return x;
}
__device__ __forceinline__ v16i mfma4(v4i a, v4i b, v16i c) {
#pragma unroll
    for (int i = 0; i < 4; i++) c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    return c;
}
template <int MODE, int NV> __global__ __launch_bounds__(256) void k(u64 *out, int iters, unsigned kk) {
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, (int)blockIdx.x, 7, 8};
    v16i c0 = {0}, c1 = {0};
    asm volatile("" : "+v"(c0)); // opaque: the VALU-only mode must not constant-fold its addends
    u64 x = threadIdx.x;
    if (MODE >= 3) c0 = mfma4(a, b, c0);
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { c0 = mfma4(a, b, c0); }
        if (MODE == 1) { x = valu_block<NV>(x, c0, kk); }
        if (MODE == 2) { c0 = mfma4(a, b, c0); x = valu_block<NV>(x, c0, kk); c0[0] = (int)x; }
        if (MODE == 4) { // as mode 3, but the scheduler is told to interleave: 1 MFMA, then a quarter of the VALU block, four times
            c1 = mfma4(a, b, c1);
            x = valu_block<NV>(x, c0, kk);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, NV * 3 / 4, 0);  // VALU
            }
            c0 = c1;
            c1[0] = (int)x;
        }
        if (MODE == 3) {
            c1 = mfma4(a, b, c1);          // next round's product, independent of the VALU block below
            x = valu_block<NV>(x, c0, kk);
            c0 = c1;
            c1[0] = (int)x;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x + (u64)c0[3] + (u64)c1[5];
}
template <int MODE, int NV> float run(u64 *d, int wgs, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NV><<<wgs, 256>>>(d, 10, 3);
    hipEventRecord(e0);
    k<MODE, NV><<<wgs, 256>>>(d, iters, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
template <int NV> void sweep(u64 *d, int iters) {
    printf("per round: 4 chained v_mfma_i32_32x32x32_i8 + %d dependent {v_mad_u64_u32, add}; ns per round per SIMD\n", NV);
    for (int occ = 1; occ <= 4; occ *= 2) { // workgroups per CU = waves per SIMD
        const int wgs = 256 * occ;
        const float f = 1e6f / iters / occ;
        const float m0 = run<0, NV>(d, wgs, iters) * f, m1 = run<1, NV>(d, wgs, iters) * f, m2 = run<2, NV>(d, wgs, iters) * f, m3 = run<3, NV>(d, wgs, iters) * f, m4 = run<4, NV>(d, wgs, iters) * f;
        printf("  waves/SIMD %d:  mfma %6.1f  valu %6.1f  serial %6.1f  pipelined %6.1f  interleaved %6.1f   (sum %6.1f, max %6.1f)\n", occ, m0, m1, m2, m3, m4, m0 + m1, m0 > m1 ? m0 : m1);
    }
}
int main() {
    u64 *d;
    hipMalloc(&d, 256 * 16 * 256 * 8);
    const int iters = 20000;
    sweep<16>(d, iters);
    sweep<64>(d, iters);
    return 0;
}
