// write_pattern.hip -- how fast does the part take the strided pass's store pattern?  A "row" is a polynomial of 32768 words (256 KiB); a workgroup of
// 256 threads writes the tile (row, t): 64 segments of SEG bytes, 4 KiB apart (segment j at word j * 512 + t * SEG / 8), 8 words per thread -- exactly
// what ntt2_kernel<0, 1, 6, 5, ...> stores with SEG = 256.  Wider segments = fewer, wider tiles of the same row.  Build: hipcc --offload-arch=gfx950 -O3
// tools/write_pattern.hip -o tools/write_pattern;  run on the GPU box: tools/write_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int SEGW, bool VEC2> __global__ __launch_bounds__(256) void wr(uint64_t *buf, unsigned tiles_per_row_log, uint64_t salt) {
    // SEGW words per segment; a tile covers 2048 words = (2048 / SEGW) segments, consecutive segments 512 words apart when SEGW <= 512
    const unsigned tile = blockIdx.x & ((1u << tiles_per_row_log) - 1), row = blockIdx.x >> tiles_per_row_log;
    uint64_t *p = buf + ((uint64_t)row << 15);
    const unsigned t = threadIdx.x;
#pragma unroll
    for (int e = 0; e < 8; e += VEC2 ? 2 : 1) {
        const unsigned q = t + 256 * e;                     // element index inside the tile, segment-major
        const unsigned w = VEC2 ? (q * 1u) : q;
        const unsigned seg = (VEC2 ? (t * 2 + 512 * (e / 2)) : w) / SEGW, off = (VEC2 ? (t * 2 + 512 * (e / 2)) : w) % SEGW;
        const unsigned idx = SEGW >= 2048 ? (tile * 2048 + (VEC2 ? (t * 2 + 512 * (e / 2)) : w)) : (seg * (32768 / (2048 / SEGW)) + tile * SEGW + off);
        if (VEC2) { ulonglong2 v; v.x = salt + idx; v.y = salt ^ idx; *reinterpret_cast<ulonglong2 *>(p + idx) = v; }
        else p[idx] = salt + idx;
    }
}

// the pattern a column-tile fusion would need: 64 segments of 64 B (8 words), 4 KiB apart, per workgroup (512 words: two per thread); READ = 1: the same as loads
template <int READ> __global__ __launch_bounds__(256) void seg64(uint64_t *buf, uint64_t salt) {
    const unsigned tile = blockIdx.x & 63, row = blockIdx.x >> 6;
    uint64_t *p = buf + ((uint64_t)row << 15);
    uint64_t acc = 0;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const unsigned q = threadIdx.x + 256 * e, idx = (q >> 3) * 512 + tile * 8 + (q & 7);
        if (READ) acc ^= p[idx]; else p[idx] = salt + idx;
    }
    if (READ && acc == 0x123456789abcdefull) p[0] = acc;
}
template <int READ> static int run64(uint64_t *buf, unsigned rows) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    seg64<READ><<<rows << 6, 256>>>(buf, 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 5; i++) seg64<READ><<<rows << 6, 256>>>(buf, i);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    std::printf("%-34s segments of    64 B,  8-byte %s: %7.1f GB/s\n", "column tile of 8 (64 x 64 B)", READ ? "loads " : "stores", 5.0 * rows * 262144.0 / (ms * 1e-3) / 1e9);
    return 0;
}

template <int SEGW, bool VEC2> static int run(uint64_t *buf, unsigned rows, const char *what) {
    const unsigned tiles_log = 4; // 16 tiles of 2048 words per row
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    wr<SEGW, VEC2><<<rows << tiles_log, 256>>>(buf, tiles_log, 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 5; i++) wr<SEGW, VEC2><<<rows << tiles_log, 256>>>(buf, tiles_log, i);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    std::printf("%-34s segments of %5d B, %s stores: %7.1f GB/s\n", what, SEGW * 8, VEC2 ? "16-byte" : " 8-byte", 5.0 * rows * 262144.0 / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    const unsigned rows = 16384; // 4 GiB
    uint64_t *buf;
    CHECK(hipMalloc(&buf, (size_t)rows << 18));
    if (run<32, false>(buf, rows, "strided pass today (64 x 256 B)")) return 1;
    if (run<32, true>(buf, rows, "the same, 16-byte stores")) return 1;
    if (run<64, false>(buf, rows, "32 x 512 B")) return 1;
    if (run<64, true>(buf, rows, "32 x 512 B")) return 1;
    if (run<128, true>(buf, rows, "16 x 1 KiB")) return 1;
    if (run<512, true>(buf, rows, "4 x 4 KiB")) return 1;
    if (run<2048, true>(buf, rows, "contiguous 16 KiB tile")) return 1;
    if (run<2048, false>(buf, rows, "contiguous 16 KiB tile")) return 1;
    if (run64<0>(buf, rows)) return 1;
    if (run64<1>(buf, rows)) return 1;
    CHECK(hipFree(buf));
    return 0;
}
