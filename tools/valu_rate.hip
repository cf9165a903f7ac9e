// Issue-rate microbenchmark for the integer VALU instructions the ORB kernels are made of (gfx950).
// 256 blocks x 1024 threads (4 waves per SIMD on every CU), each wave runs REPS x 64 independent instructions of one kind.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REPS 4096
#define OP16(S) S S S S S S S S S S S S S S S S
template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned *out) {
    unsigned a = threadIdx.x, b = blockIdx.x + 1, c = 7, d = 3;
    for (int i = 0; i < REPS; ++i) {
        if (KIND == 0) { OP16(asm volatile("v_xor_b32 %0, %1, %0\n v_xor_b32 %2, %1, %2\n v_xor_b32 %3, %1, %3\n v_xor_b32 %1, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 1) { OP16(asm volatile("v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %2, %1, %2\n v_bcnt_u32_b32 %3, %1, %3\n v_bcnt_u32_b32 %1, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 2) { OP16(asm volatile("v_min_u32 %0, %1, %0\n v_max_u32 %2, %1, %2\n v_min_u32 %3, %1, %3\n v_max_u32 %1, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 3) { OP16(asm volatile("v_add_u32 %0, %1, %0\n v_add_u32 %2, %1, %2\n v_add_u32 %3, %1, %3\n v_add_u32 %1, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 4) { OP16(asm volatile("v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %2, %1, %2, %2\n v_fma_f32 %3, %1, %3, %3\n v_fma_f32 %1, %0, %1, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 5) { OP16(asm volatile("v_lshl_or_b32 %0, %1, 3, %0\n v_lshl_or_b32 %2, %1, 3, %2\n v_lshl_or_b32 %3, %1, 3, %3\n v_lshl_or_b32 %1, %0, 3, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 6) { OP16(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_gt_u32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
        if (KIND == 7) { OP16(asm volatile("v_mad_u32_u24 %0, %1, %0, %0\n v_mad_u32_u24 %2, %1, %2, %2\n v_mad_u32_u24 %3, %1, %3, %3\n v_mad_u32_u24 %1, %0, %1, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    }
    if ((a ^ b ^ c ^ d) == 0x12345678u) out[0] = a;
}
template <int KIND> void run(const char *name, unsigned *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, 1024>>>(out); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<256, 1024>>>(out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x REPS x 64 instructions
    const double inst_per_simd = 4.0 * REPS * 64;
    printf("%-16s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4 GHz, %.2f @2.1 GHz)\n", name, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4, ms * 1e6 / inst_per_simd * 2.1);
}
int main() {
    unsigned *out; hipMalloc(&out, 4);
    run<0>("v_xor_b32", out); run<1>("v_bcnt_u32_b32", out); run<2>("v_min/max_u32", out); run<3>("v_add_u32", out);
    run<4>("v_fma_f32", out); run<5>("v_lshl_or_b32", out); run<6>("v_cmp+v_cndmask", out); run<7>("v_mad_u32_u24", out);
    return 0;
}
