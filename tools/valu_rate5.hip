// Fifth issue-rate table (gfx950): v_cndmask_b32 (9.8 ns in valu_rate2 with a constant VCC) against its alternatives.
//   hipcc --offload-arch=gfx950 -O3 -w tools/valu_rate5.hip -o tools/bin/valu_rate5 && tools/bin/valu_rate5
#include <hip/hip_runtime.h>
#include <cstdio>
#define REPS 2048
#define OP16(S) S S S S S S S S S S S S S S S S
#define Q4(I) asm volatile(I : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "s"(sg), "s"(sm) : "vcc");

template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned *out, unsigned sg, unsigned long long sm) {
    unsigned a = threadIdx.x, b = blockIdx.x + 1, c = 7, d = 3, e = threadIdx.x * 4;
    asm volatile("s_mov_b64 vcc, %0" : : "s"(sm) : "vcc");
    for (int i = 0; i < REPS; ++i) {
        if (KIND == 0) { OP16(Q4("v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %2, %1, %2, vcc\n v_cndmask_b32 %3, %1, %3, vcc\n v_cndmask_b32 %1, %0, %1, vcc")) }
        if (KIND == 1) { OP16(Q4("v_cndmask_b32_e64 %0, %1, %0, %6\n v_cndmask_b32_e64 %2, %1, %2, %6\n v_cndmask_b32_e64 %3, %1, %3, %6\n v_cndmask_b32_e64 %1, %0, %1, %6")) }
        if (KIND == 2) { OP16(Q4("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %1, %2, vcc\n v_cmp_gt_u32 vcc, %3, %1\n v_cndmask_b32 %0, %1, %3, vcc")) }
        if (KIND == 3) { OP16(Q4("v_cndmask_b32 %0, 0, %0, vcc\n v_cndmask_b32 %2, 0, %2, vcc\n v_cndmask_b32 %3, 0, %3, vcc\n v_cndmask_b32 %1, 0, %1, vcc")) }
        if (KIND == 4) { OP16(Q4("v_bfi_b32 %0, %4, %1, %0\n v_bfi_b32 %2, %4, %1, %2\n v_bfi_b32 %3, %4, %1, %3\n v_bfi_b32 %1, %4, %0, %1")) }
        if (KIND == 5) { OP16(Q4("v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %2, %1\n v_cmp_gt_u32 vcc, %3, %1\n v_cndmask_b32 %0, %1, %3, vcc")) }
        if (KIND == 6) { OP16(Q4("v_cmp_gt_u32 vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_cmp_gt_u32 vcc, %3, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc")) }
        if (KIND == 8) { OP16(Q4("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %1, %2, vcc\n v_cndmask_b32 %3, %1, %3, vcc\n v_cndmask_b32 %0, %1, %3, vcc")) }
        if (KIND == 9) { OP16(asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %2, %1, %2, s[20:21]\n v_cndmask_b32_e64 %3, %1, %3, s[20:21]\n v_cndmask_b32_e64 %0, %1, %3, s[20:21]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21");) }
        if (KIND == 10) { OP16(Q4("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %1, %2, vcc\n v_add_u32 %3, %1, %3\n v_cndmask_b32 %0, %1, %3, vcc")) }
        if (KIND == 11) { OP16(Q4("v_cmp_gt_u32 vcc, %0, %1\n v_add_u32 %3, %1, %3\n v_add_u32 %2, %1, %2\n v_cndmask_b32 %0, %1, %3, vcc")) }
        if (KIND == 7) { OP16(Q4("v_cmp_gt_u32 vcc, %0, %1\n v_subb_co_u32 %2, vcc, %2, %2, vcc\n v_cmp_gt_u32 vcc, %3, %1\n v_subb_co_u32 %0, vcc, %0, %0, vcc")) }
    }
    if (a + b + c + d == 0x12345678) out[0] = a;
}
template <int KIND>
static void run(const char *name, unsigned *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, 1024>>>(out, 12345u, 0x5555aaaa0f0f3333ull); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<256, 1024>>>(out, 12345u, 0x5555aaaa0f0f3333ull); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = 4.0 * REPS * 16 * 4;
    printf("%-40s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4 GHz)\n", name, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    fflush(stdout);
}
int main() {
    unsigned *out; hipMalloc(&out, 4);
    run<0>("v_cndmask vcc (constant mask)", out); run<1>("v_cndmask_e64 sgpr pair", out); run<3>("v_cndmask 0, v, vcc", out);
    run<2>("v_cmp + v_cndmask pairs", out); run<5>("3 v_cmp + 1 v_cndmask", out); run<4>("v_bfi_b32 (mask in a VGPR)", out);
    run<8>("v_cmp + 3 v_cndmask vcc", out); run<9>("v_cmp_e64 sgpr + 3 v_cndmask_e64", out); run<10>("v_cmp, cndmask, add, cndmask", out); run<11>("v_cmp, add, add, cndmask", out);
    run<6>("v_cmp + v_addc pairs", out); run<7>("v_cmp + v_subb pairs", out);
    return 0;
}
