// mfma_probe.hip -- lane/operand layout check of v_mfma_i32_32x32x32_i8 on gfx950 (development tool for behz.hip).
// D[m][n] = sum_k A[m][k] * B[k][n], signed int8 inputs.  Assumed layout (ck_tile WarpGemmAttributeMfmaImpl_i32_32x32x32_i8):
//   A: lane l holds row m = l % 32, bytes k = 16 * (l / 32) + 0..15      B: lane l holds column n = l % 32, same k range
//   D: lane l holds column n = l % 32, rows m = 8 * (r / 4) + 4 * (l / 32) + r % 4 for r = 0..15
// build + run: hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void probe(const int8_t *A, const int8_t *B, int *D) {
    const int l = threadIdx.x;
    v4i a = *reinterpret_cast<const v4i *>(A + (l % 32) * 32 + 16 * (l / 32));   // A row-major [32][32]
    v4i b;
    int8_t tmp[16];
    for (int i = 0; i < 16; i++) tmp[i] = B[(16 * (l / 32) + i) * 32 + (l % 32)]; // B row-major [k][n]
    b = *reinterpret_cast<v4i *>(tmp);
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[(8 * (r / 4) + 4 * (l / 32) + r % 4) * 32 + (l % 32)] = c[r];
}
int main() {
    int8_t hA[1024], hB[1024];
    int hD[1024], ref[1024];
    srand(1);
    for (int i = 0; i < 1024; i++) { hA[i] = (int8_t)(rand() % 256 - 128); hB[i] = (int8_t)(rand() % 256 - 128); }
    for (int m = 0; m < 32; m++)
        for (int n = 0; n < 32; n++) {
            int s = 0;
            for (int k = 0; k < 32; k++) s += (int)hA[m * 32 + k] * (int)hB[k * 32 + n];
            ref[m * 32 + n] = s;
        }
    int8_t *dA, *dB;
    int *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; i++) bad += hD[i] != ref[i];
    printf("mfma_i32_32x32x32_i8 layout check: %d mismatches of 1024%s\n", bad, bad ? "" : "  (layout as assumed)");
    return bad != 0;
}
