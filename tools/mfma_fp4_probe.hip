// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (e2m1) operands on gfx950, exact integer data (the k_match fp4 experiment):
//   * lane (r = l & 31, h = l >> 5) supplies 32 K elements as the 32 nibbles of its first four operand registers;
//   * A nibble 0b0100 (2.0) x B nibble 0b1110 (-4.0) = -8 per common bit, times the block scales 2^(sa-127) * 2^(sb-127);
//   * D = C + sum, C/D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// Build + run:  hipcc --offload-arch=gfx950 -O2 tools/mfma_fp4_probe.hip -o tools/bin/mfma_fp4_probe && tools/bin/mfma_fp4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void expand(unsigned x, unsigned nib, int *out) {   // bit 4j+k of x -> nibble j of out[k]
    for (int k = 0; k < 4; ++k) out[k] = (int)(((x >> k) & 0x11111111u) * nib);
}
__global__ void k(const unsigned *abits, const unsigned *bbits, float *d, int sa, int sb) {
    const int l = threadIdx.x;
    v8i A = {0, 0, 0, 0, 0, 0, 0, 0}, B = {0, 0, 0, 0, 0, 0, 0, 0};
    int t[4];
    expand(abits[l], 0x4u, t); for (int i = 0; i < 4; ++i) A[i] = t[i];
    expand(bbits[l], 0xEu, t); for (int i = 0; i < 4; ++i) B[i] = t[i];
    v16f c;
    for (int i = 0; i < 16; ++i) c[i] = 3000000.f + (float)(l * 16 + i);
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) d[l * 16 + i] = c[i];
}
int main() {
    unsigned ha[64], hb[64], *da, *db; float hd[1024], *dd;
    srand(7);
    for (int i = 0; i < 64; ++i) { ha[i] = (unsigned)rand() ^ ((unsigned)rand() << 16); hb[i] = (unsigned)rand() ^ ((unsigned)rand() << 16); }
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 4096);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    const int cases[4][2] = {{127, 127}, {137, 127}, {127, 137}, {132, 132}};
    int bad_total = 0;
    for (int cs = 0; cs < 4; ++cs) {
        const int sa = cases[cs][0], sb = cases[cs][1];
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd, sa | (99 << 8), sb | (55 << 8));   // upper bytes must be ignored (opsel 0)
        if (hipMemcpy(hd, dd, 4096, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 2; }
        const double scale = ldexp(1.0, sa - 127 + sb - 127);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int reg = 0; reg < 16; ++reg) {
                const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
                int dot = 0;
                for (int h = 0; h < 2; ++h) dot += __builtin_popcount(ha[row + 32 * h] & hb[col + 32 * h]);
                const double want = 3000000.0 + (l * 16 + reg) - 8.0 * scale * dot;
                if ((double)hd[l * 16 + reg] != want) { if (bad < 4) printf("  case %d lane %d reg %d: got %.1f want %.1f (dot %d)\n", cs, l, reg, hd[l * 16 + reg], want, dot); ++bad; }
            }
        printf("scale_a %d scale_b %d: %d of 1024 differ\n", sa, sb, bad);
        bad_total += bad;
    }
    printf(bad_total ? "PROBE FAILED\n" : "PROBE OK\n");
    return bad_total != 0;
}
