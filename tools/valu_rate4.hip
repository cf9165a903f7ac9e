// Fourth issue-rate table (gfx950): does the OPERAND KIND change the rate of a 32-bit-encoded VOP2/VOP1 instruction?
// (valu_rate3 showed v_xor_b32 with an SGPR source at 1.84 ns against 1.21 ns with two VGPRs.)  Same harness as valu_rate2.hip.
//   hipcc --offload-arch=gfx950 -O3 -w tools/valu_rate4.hip -o tools/bin/valu_rate4 && tools/bin/valu_rate4
#include <hip/hip_runtime.h>
#include <cstdio>
#define REPS 2048
#define OP16(S) S S S S S S S S S S S S S S S S
#define Q4(I) asm volatile(I : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "s"(sg) : "vcc");

template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned *out, unsigned sg) {
    unsigned a = threadIdx.x, b = blockIdx.x + 1, c = 7, d = 3, e = threadIdx.x * 4;
    for (int i = 0; i < REPS; ++i) {
        if (KIND == 0) { OP16(Q4("v_add_u32 %0, %1, %0\n v_add_u32 %2, %1, %2\n v_add_u32 %3, %1, %3\n v_add_u32 %1, %0, %1")) }
        if (KIND == 1) { OP16(Q4("v_add_u32 %0, 3, %0\n v_add_u32 %2, 5, %2\n v_add_u32 %3, 7, %3\n v_add_u32 %1, 9, %1")) }
        if (KIND == 2) { OP16(Q4("v_add_u32 %0, 0x12345, %0\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3\n v_add_u32 %1, 0x12345, %1")) }
        if (KIND == 3) { OP16(Q4("v_add_u32 %0, %5, %0\n v_add_u32 %2, %5, %2\n v_add_u32 %3, %5, %3\n v_add_u32 %1, %5, %1")) }
        if (KIND == 4) { OP16(Q4("v_lshlrev_b32 %0, %4, %0\n v_lshrrev_b32 %2, %4, %2\n v_lshlrev_b32 %3, %4, %3\n v_lshrrev_b32 %1, %4, %1")) }
        if (KIND == 5) { OP16(Q4("v_and_b32 %0, 0xff00ff, %0\n v_or_b32 %2, 0xff00ff, %2\n v_and_b32 %3, 0xff00ff, %3\n v_or_b32 %1, 0xff00ff, %1")) }
        if (KIND == 6) { OP16(Q4("v_add_f32 %0, %1, %0\n v_add_f32 %2, %1, %2\n v_add_f32 %3, %1, %3\n v_add_f32 %1, %0, %1")) }
        if (KIND == 7) { OP16(Q4("v_mul_f32 %0, %1, %0\n v_mul_f32 %2, %1, %2\n v_mul_f32 %3, %1, %3\n v_mul_f32 %1, %0, %1")) }
        if (KIND == 8) { OP16(Q4("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %2, %1, %3\n v_fmac_f32 %3, %1, %0\n v_fmac_f32 %1, %0, %2")) }
        if (KIND == 9) { OP16(Q4("v_add_f32 %0, 0x4b400000, %0\n v_add_f32 %2, 0x4b400000, %2\n v_add_f32 %3, 0x4b400000, %3\n v_add_f32 %1, 0x4b400000, %1")) }
        if (KIND == 10) { OP16(Q4("v_cvt_f32_u32 %0, %1\n v_cvt_f32_u32 %2, %3\n v_cvt_f32_u32 %3, %0\n v_cvt_f32_u32 %1, %2")) }
        if (KIND == 11) { OP16(Q4("v_rndne_f32 %0, %1\n v_rndne_f32 %2, %3\n v_rndne_f32 %3, %0\n v_rndne_f32 %1, %2")) }
        if (KIND == 12) { OP16(Q4("v_min_f32 %0, %1, %0\n v_max_f32 %2, %1, %2\n v_min_f32 %3, %1, %3\n v_max_f32 %1, %0, %1")) }
        if (KIND == 13) { OP16(Q4("v_mul_u32_u24 %0, %1, %0\n v_mul_u32_u24 %2, %1, %2\n v_mul_u32_u24 %3, %1, %3\n v_mul_u32_u24 %1, %0, %1")) }
        if (KIND == 14) { OP16(Q4("v_mul_u32_u24 %0, 55, %0\n v_mul_u32_u24 %2, 49, %2\n v_mul_u32_u24 %3, 34, %3\n v_mul_u32_u24 %1, 18, %1")) }
        if (KIND == 15) { OP16(Q4("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n v_mov_b32 %1, %2")) }
        if (KIND == 16) { OP16(Q4("v_lshlrev_b32 %0, 3, %0\n v_lshrrev_b32 %2, 5, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshrrev_b32 %1, 8, %1")) }
        if (KIND == 17) { OP16(Q4("v_cvt_i32_f32 %0, %1\n v_cvt_i32_f32 %2, %3\n v_cvt_i32_f32 %3, %0\n v_cvt_i32_f32 %1, %2")) }
        if (KIND == 18) { OP16(Q4("v_max_u16 %0, %5, %0\n v_min_u16 %2, %5, %2\n v_max_i16 %3, %5, %3\n v_min_i16 %1, %5, %1")) }
        if (KIND == 19) { OP16(Q4("v_sub_u16 %0, %1, %0\n v_add_u16 %2, %1, %2\n v_sub_u16 %3, %1, %3\n v_add_u16 %1, %0, %1")) }
        if (KIND == 20) { OP16(Q4("v_xor_b32 %0, 0xff, %0\n v_xor_b32 %2, 0xff, %2\n v_xor_b32 %3, 0xff, %3\n v_xor_b32 %1, 0xff, %1")) }
        if (KIND == 21) { OP16(Q4("v_mul_f32 %0, 0x37800000, %0\n v_mul_f32 %2, 0x37800000, %2\n v_mul_f32 %3, 0x37800000, %3\n v_mul_f32 %1, 0x37800000, %1")) }
        if (KIND == 22) { OP16(Q4("v_mad_u32_u24 %0, %1, %0, %2\n v_mad_u32_u24 %2, %1, %2, %3\n v_mad_u32_u24 %3, %1, %3, %0\n v_mad_u32_u24 %1, %0, %1, %2")) }
        if (KIND == 23) { OP16(Q4("v_lshl_add_u32 %0, %1, 1, %2\n v_lshl_add_u32 %2, %1, 1, %3\n v_lshl_add_u32 %3, %1, 1, %0\n v_lshl_add_u32 %1, %0, 1, %2")) }
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
    run<0>("v_add_u32 v,v", out); run<1>("v_add_u32 inline const", out); run<2>("v_add_u32 literal", out); run<3>("v_add_u32 sgpr", out);
    run<4>("v_lshl/lshr vgpr amount", out); run<16>("v_lshl/lshr inline amount", out); run<5>("v_and/or literal", out); run<20>("v_xor inline 0xff->literal?", out);
    run<6>("v_add_f32 v,v", out); run<9>("v_add_f32 literal", out); run<7>("v_mul_f32 v,v", out); run<21>("v_mul_f32 literal", out); run<8>("v_fmac_f32", out);
    run<10>("v_cvt_f32_u32", out); run<17>("v_cvt_i32_f32", out); run<11>("v_rndne_f32", out); run<12>("v_min/max_f32", out);
    run<13>("v_mul_u32_u24 v,v", out); run<14>("v_mul_u32_u24 inline", out); run<22>("v_mad_u32_u24", out); run<23>("v_lshl_add_u32", out);
    run<15>("v_mov_b32", out); run<18>("v_min/max_16 sgpr", out); run<19>("v_add/sub_u16", out);
    return 0;
}
