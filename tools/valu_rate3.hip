// Third issue-rate table (gfx950): the Hamming inner loop's instructions (v_bcnt_u32_b32 with accumulate, v_xor with an
// SGPR operand, the xor+bcnt pair) and the top-2 bookkeeping.  Same harness as valu_rate2.hip.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate3.hip -o tools/bin/valu_rate3 && tools/bin/valu_rate3
#include <hip/hip_runtime.h>
#include <cstdio>
#define REPS 2048
#define OP16(S) S S S S S S S S S S S S S S S S
#define Q4(I) asm volatile(I : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "s"(sg) : "vcc");

template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned *out, unsigned sg) {
    unsigned a = threadIdx.x, b = blockIdx.x + 1, c = 7, d = 3, e = threadIdx.x * 4;
    for (int i = 0; i < REPS; ++i) {
        if (KIND == 0) { OP16(Q4("v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %2, %1, %2\n v_bcnt_u32_b32 %3, %1, %3\n v_bcnt_u32_b32 %1, %0, %1")) }
        if (KIND == 1) { OP16(Q4("v_xor_b32 %0, %5, %0\n v_xor_b32 %2, %5, %2\n v_xor_b32 %3, %5, %3\n v_xor_b32 %1, %5, %1")) }
        if (KIND == 2) { OP16(Q4("v_xor_b32 %0, %4, %1\n v_bcnt_u32_b32 %2, %0, %2\n v_xor_b32 %3, %4, %1\n v_bcnt_u32_b32 %2, %3, %2")) }
        if (KIND == 3) { OP16(Q4("v_xor_b32 %0, %5, %1\n v_bcnt_u32_b32 %2, %0, %2\n v_xor_b32 %3, %5, %1\n v_bcnt_u32_b32 %2, %3, %2")) }
        if (KIND == 4) { OP16(Q4("v_bcnt_u32_b32 %0, %1, 0\n v_bcnt_u32_b32 %2, %1, 0\n v_bcnt_u32_b32 %3, %1, 0\n v_bcnt_u32_b32 %1, %0, 0")) }
        if (KIND == 5) { OP16(Q4("v_med3_u32 %0, %1, %0, %2\n v_min_u32 %2, %1, %2\n v_lshl_or_b32 %3, %1, 20, %5\n v_min_u32 %1, %0, %1")) }
        if (KIND == 6) { OP16(Q4("v_and_b32 %0, %5, %1\n v_bcnt_u32_b32 %2, %0, %2\n v_and_b32 %3, %5, %1\n v_bcnt_u32_b32 %2, %3, %2")) }
    }
    if (a + b + c + d == 0x12345678) out[0] = a;
}
template <int KIND>
static void run(const char *name, unsigned *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, 1024>>>(out, 12345u); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<256, 1024>>>(out, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = 4.0 * REPS * 16 * 4;
    printf("%-36s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4 GHz)\n", name, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    fflush(stdout);
}
int main() {
    unsigned *out; hipMalloc(&out, 4);
    run<0>("v_bcnt_u32_b32 (accumulate)", out); run<4>("v_bcnt_u32_b32 (+0)", out); run<1>("v_xor_b32 sgpr operand", out);
    run<2>("xor(v,v) + bcnt pairs", out); run<3>("xor(s,v) + bcnt pairs", out); run<6>("and(s,v) + bcnt pairs", out);
    run<5>("med3/min/lshl_or/min", out);
    return 0;
}
