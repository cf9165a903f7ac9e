// Sixth issue-rate table (gfx950): the 16-bit min / max forms the one-pass FAST corner test is made of (v_min3_i16 / v_max3_i16
// were not in the earlier tables), next to the v_cmp + v_addc pairs of the ring-mask test.
//   hipcc --offload-arch=gfx950 -O3 -w tools/valu_rate6.hip -o tools/bin/valu_rate6 && tools/bin/valu_rate6
#include <hip/hip_runtime.h>
#include <cstdio>
#define REPS 2048
#define OP16(S) S S S S S S S S S S S S S S S S
#define Q4(I) asm volatile(I : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "s"(sg) : "vcc");

template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned *out, unsigned sg) {
    unsigned a = threadIdx.x, b = blockIdx.x + 1, c = 7, d = 3, e = threadIdx.x * 4;
    for (int i = 0; i < REPS; ++i) {
        if (KIND == 0) { OP16(Q4("v_min3_i16 %0, %1, %4, %0\n v_min3_i16 %2, %1, %4, %2\n v_min3_i16 %3, %1, %4, %3\n v_min3_i16 %1, %0, %4, %1")) }
        if (KIND == 1) { OP16(Q4("v_max3_i16 %0, %1, %4, %0\n v_max3_i16 %2, %1, %4, %2\n v_max3_i16 %3, %1, %4, %3\n v_max3_i16 %1, %0, %4, %1")) }
        if (KIND == 2) { OP16(Q4("v_min_i16 %0, %1, %0\n v_min_i16 %2, %1, %2\n v_min_i16 %3, %1, %3\n v_min_i16 %1, %0, %1")) }
        if (KIND == 3) { OP16(Q4("v_sub_u16 %0, %1, %0\n v_sub_u16 %2, %1, %2\n v_sub_u16 %3, %1, %3\n v_sub_u16 %1, %0, %1")) }
        if (KIND == 4) { OP16(Q4("v_xor_b32 %0, %1, %0\n v_xor_b32 %2, %1, %2\n v_xor_b32 %3, %1, %3\n v_xor_b32 %1, %0, %1")) }
        if (KIND == 5) { OP16(Q4("v_min3_i32 %0, %1, %4, %0\n v_min3_i32 %2, %1, %4, %2\n v_min3_i32 %3, %1, %4, %3\n v_min3_i32 %1, %0, %4, %1")) }
        if (KIND == 6) { OP16(Q4("v_cmp_gt_i32 vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_cmp_gt_i32 vcc, %3, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc")) }
        if (KIND == 7) { OP16(Q4("v_min_i16 %0, %1, %0\n v_min_i16 %0, %4, %0\n v_min_i16 %0, %2, %0\n v_min_i16 %0, %3, %0")) }   // one dependent chain
        if (KIND == 8) { OP16(Q4("v_min3_i16 %0, %1, %4, %0\n v_min3_i16 %0, %2, %4, %0\n v_min3_i16 %0, %3, %4, %0\n v_min3_i16 %0, %1, %2, %0")) }   // one dependent chain
        if (KIND == 9) { OP16(Q4("v_pk_min_i16 %0, %1, %0\n v_pk_min_i16 %2, %1, %2\n v_pk_min_i16 %3, %1, %3\n v_pk_min_i16 %1, %0, %1")) }
        if (KIND == 10) { OP16(Q4("v_ashrrev_i16 %0, 15, %0\n v_ashrrev_i16 %2, 15, %2\n v_ashrrev_i16 %3, 15, %3\n v_ashrrev_i16 %1, 15, %1")) }
    }
    if (a + b + c + d == 0x12345678) out[0] = a;
}
template <int KIND>
static void run(const char *name, unsigned *out, int threads) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, threads>>>(out, 12345u); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<256, threads>>>(out, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (threads / 256.0) * REPS * 16 * 4;
    printf("%-34s %4d thr/CU %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4 GHz)\n", name, threads, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    fflush(stdout);
}
int main() {
    unsigned *out; hipMalloc(&out, 4);
    for (int threads : {1024, 256}) {   // 4 waves and 1 wave per SIMD
        run<4>("v_xor_b32", out, threads); run<2>("v_min_i16", out, threads); run<3>("v_sub_u16", out, threads); run<10>("v_ashrrev_i16", out, threads);
        run<0>("v_min3_i16", out, threads); run<1>("v_max3_i16", out, threads); run<5>("v_min3_i32", out, threads); run<9>("v_pk_min_i16", out, threads);
        run<6>("v_cmp + v_addc pairs", out, threads); run<7>("v_min_i16 dependent chain", out, threads); run<8>("v_min3_i16 dependent chain", out, threads);
    }
    return 0;
}
