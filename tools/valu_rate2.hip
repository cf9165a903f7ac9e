// Second issue-rate table (gfx950): the candidate instructions for a packed / DPP formulation of FAST and rBRIEF.
// 256 blocks x 1024 threads (4 waves per SIMD on every CU), each wave runs REPS x 64 instructions of one kind on four
// independent dependency chains.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate2.hip -o /tmp/valu_rate2 && /tmp/valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#define REPS 2048
#define OP16(S) S S S S S S S S S S S S S S S S
#define Q4(I) asm volatile(I : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");

template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned *out) {
    __shared__ unsigned lds[2048];
    unsigned a = threadIdx.x, b = blockIdx.x + 1, c = 7, d = 3, e = threadIdx.x * 4;
    lds[threadIdx.x] = a; lds[threadIdx.x + 1024] = b;
    __syncthreads();
    for (int i = 0; i < REPS; ++i) {
        if (KIND == 0) { OP16(Q4("v_xor_b32 %0, %1, %0\n v_xor_b32 %2, %1, %2\n v_xor_b32 %3, %1, %3\n v_xor_b32 %1, %0, %1")) }
        if (KIND == 1) { OP16(Q4("v_pk_max_u16 %0, %1, %0\n v_pk_min_u16 %2, %1, %2\n v_pk_max_u16 %3, %1, %3\n v_pk_min_u16 %1, %0, %1")) }
        if (KIND == 2) { OP16(Q4("v_pk_sub_i16 %0, %1, %0\n v_pk_add_u16 %2, %1, %2\n v_pk_sub_i16 %3, %1, %3\n v_pk_add_u16 %1, %0, %1")) }
        if (KIND == 3) { OP16(Q4("v_perm_b32 %0, %1, %0, %4\n v_perm_b32 %2, %1, %2, %4\n v_perm_b32 %3, %1, %3, %4\n v_perm_b32 %1, %0, %1, %4")) }
        if (KIND == 4) { OP16(Q4("v_alignbyte_b32 %0, %1, %0, 3\n v_alignbyte_b32 %2, %1, %2, 1\n v_alignbyte_b32 %3, %1, %3, 3\n v_alignbyte_b32 %1, %0, %1, 1")) }
        if (KIND == 5) { OP16(Q4("v_max3_u32 %0, %1, %0, %2\n v_min3_u32 %2, %1, %2, %3\n v_max3_u32 %3, %1, %3, %0\n v_min3_u32 %1, %0, %1, %2")) }
        if (KIND == 6) { OP16(Q4("v_med3_i32 %0, %1, %0, %2\n v_med3_i32 %2, %1, %2, %3\n v_med3_i32 %3, %1, %3, %0\n v_med3_i32 %1, %0, %1, %2")) }
        if (KIND == 7) { OP16(Q4("v_max_i16 %0, %1, %0\n v_min_i16 %2, %1, %2\n v_max_u16 %3, %1, %3\n v_min_u16 %1, %0, %1")) }
        if (KIND == 8) { OP16(Q4("v_max_i32 %0, %1, %0\n v_min_i32 %2, %1, %2\n v_max_i32 %3, %1, %3\n v_min_i32 %1, %0, %1")) }
        if (KIND == 9) { OP16(Q4("v_sub_u32 %0, %1, %0\n v_sub_u32 %2, %1, %2\n v_subrev_u32 %3, %1, %3\n v_sub_u32 %1, %0, %1")) }
        if (KIND == 10) { OP16(Q4("v_and_b32 %0, %1, %0\n v_or_b32 %2, %1, %2\n v_and_b32 %3, %1, %3\n v_or_b32 %1, %0, %1")) }
        if (KIND == 11) { OP16(Q4("v_and_or_b32 %0, %1, %0, %2\n v_and_or_b32 %2, %1, %2, %3\n v_and_or_b32 %3, %1, %3, %0\n v_and_or_b32 %1, %0, %1, %2")) }
        if (KIND == 12) { OP16(Q4("v_bfe_u32 %0, %1, 8, 8\n v_bfe_u32 %2, %1, 16, 8\n v_bfe_u32 %3, %1, 8, 8\n v_bfe_u32 %1, %0, 8, 8")) }
        if (KIND == 13) { OP16(Q4("v_lshrrev_b32 %0, 8, %1\n v_lshlrev_b32 %2, 3, %1\n v_lshrrev_b32 %3, 5, %1\n v_lshlrev_b32 %1, 1, %0")) }
        if (KIND == 14) { OP16(Q4("v_mov_b32_dpp %0, %1 row_shr:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %1 row_shl:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf")) }
        if (KIND == 15) { OP16(Q4("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf")) }
        if (KIND == 16) { OP16(Q4("v_max_u32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %2, %1, %2 row_shl:3 row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %3, %1, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %1, %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf")) }
        if (KIND == 17) { OP16(Q4("v_xor_b32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf\n v_xor_b32_dpp %2, %1, %2 row_shl:3 row_mask:0xf bank_mask:0xf\n v_xor_b32_dpp %3, %1, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_xor_b32_dpp %1, %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf")) }
        if (KIND == 18) { OP16(Q4("v_sub_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n v_sub_u32_sdwa %2, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_0\n v_sub_u32_sdwa %3, %1, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_0\n v_sub_u32_sdwa %1, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")) }
        if (KIND == 19) { OP16(Q4("v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %1, %0\n v_cmp_lt_u32 vcc, %3, %2")) }
        if (KIND == 20) { OP16(asm volatile("v_cmp_gt_u32 s[20:21], %0, %1\n v_cmp_gt_u32 s[22:23], %2, %3\n s_and_b64 s[24:25], s[20:21], s[22:23]\n v_cmp_lt_u32 s[20:21], %1, %0\n v_cmp_lt_u32 s[22:23], %3, %2\n s_or_b64 s[24:25], s[20:21], s[22:23]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (KIND == 21) { OP16(Q4("v_cmp_gt_u16_sdwa vcc, %0, %1 src0_sel:BYTE_1 src1_sel:BYTE_0\n v_cmp_gt_u16_sdwa vcc, %2, %3 src0_sel:BYTE_2 src1_sel:BYTE_0\n v_cmp_lt_u16_sdwa vcc, %1, %0 src0_sel:BYTE_3 src1_sel:BYTE_0\n v_cmp_lt_u16_sdwa vcc, %3, %2 src0_sel:BYTE_1 src1_sel:BYTE_0")) }
        if (KIND == 22) { OP16(Q4("v_mbcnt_lo_u32_b32 %0, %1, %0\n v_mbcnt_hi_u32_b32 %2, %1, %2\n v_mbcnt_lo_u32_b32 %3, %1, %3\n v_mbcnt_hi_u32_b32 %1, %0, %1")) }
        if (KIND == 23) { OP16(Q4("v_sad_u8 %0, %1, %0, %2\n v_sad_u8 %2, %1, %2, %3\n v_sad_u8 %3, %1, %3, %0\n v_sad_u8 %1, %0, %1, %2")) }
        if (KIND == 24) { OP16(Q4("v_dot4_u32_u8 %0, %1, %0, %2\n v_dot4_u32_u8 %2, %1, %2, %3\n v_dot4_u32_u8 %3, %1, %3, %0\n v_dot4_u32_u8 %1, %0, %1, %2")) }
        if (KIND == 25) { OP16(Q4("v_bfi_b32 %0, %1, %0, %2\n v_bfi_b32 %2, %1, %2, %3\n v_bfi_b32 %3, %1, %3, %0\n v_bfi_b32 %1, %0, %1, %2")) }
        if (KIND == 26) { OP16(Q4("v_pk_lshrrev_b16 %0, 3, %1\n v_pk_lshlrev_b16 %2, 3, %1\n v_pk_ashrrev_i16 %3, 15, %1\n v_pk_lshrrev_b16 %1, 1, %0")) }
        if (KIND == 27) { OP16(Q4("v_add3_u32 %0, %1, %0, %2\n v_add3_u32 %2, %1, %2, %3\n v_add3_u32 %3, %1, %3, %0\n v_add3_u32 %1, %0, %1, %2")) }
        if (KIND == 28) { OP16(Q4("v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %2, %1, %2, vcc\n v_cndmask_b32 %3, %1, %3, vcc\n v_cndmask_b32 %1, %0, %1, vcc")) }
        if (KIND == 29) { OP16(Q4("v_alignbit_b32 %0, %1, %0, 16\n v_alignbit_b32 %2, %1, %2, 16\n v_alignbit_b32 %3, %1, %3, 16\n v_alignbit_b32 %1, %0, %1, 16")) }
        if (KIND == 30) { OP16(Q4("v_pk_mad_u16 %0, %1, %0, %2\n v_pk_mad_u16 %2, %1, %2, %3\n v_pk_mad_u16 %3, %1, %3, %0\n v_pk_mad_u16 %1, %0, %1, %2")) }
        if (KIND == 31) { OP16(Q4("v_or3_b32 %0, %1, %0, %2\n v_or3_b32 %2, %1, %2, %3\n v_or3_b32 %3, %1, %3, %0\n v_or3_b32 %1, %0, %1, %2")) }
        // LDS instruction issue (addresses in e: conflict-free dword per lane; byte reads from the same dwords)
        if (KIND == 40) { OP16(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(e));) }
        if (KIND == 41) { OP16(asm volatile("ds_read_u8 %0, %4\n ds_read_u8 %1, %4 offset:257\n ds_read_u8 %2, %4 offset:514\n ds_read_u8 %3, %4 offset:771\n s_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(e));) }
        if (KIND == 42) { OP16(asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:512\n s_waitcnt lgkmcnt(0)" : "=v"(*(unsigned long long *)&a), "=v"(*(unsigned long long *)&c) : "v"(e * 2));) }
        if (KIND == 43) { OP16(asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (KIND == 44) { OP16(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(*(uint4 *)&lds[0]) : "v"(e * 4) : "memory");) }
    }
    if ((a ^ b ^ c ^ d) == 0x12345678u) out[0] = a;
}
template <int KIND> void run(const char *name, unsigned *out, double per_group = 4.0) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, 1024>>>(out); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<256, 1024>>>(out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x REPS x 16 groups x per_group instructions
    const double inst_per_simd = 4.0 * REPS * 16 * per_group;
    printf("%-28s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4 GHz)\n", name, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    fflush(stdout);
}
int main() {
    unsigned *out; hipMalloc(&out, 4);
    run<0>("v_xor_b32", out); run<1>("v_pk_max/min_u16", out); run<2>("v_pk_sub_i16/add_u16", out); run<3>("v_perm_b32", out);
    run<4>("v_alignbyte_b32", out); run<5>("v_max3/min3_u32", out); run<6>("v_med3_i32", out); run<7>("v_max/min_i16/u16", out);
    run<8>("v_max/min_i32", out); run<9>("v_sub_u32", out); run<10>("v_and/or_b32", out); run<11>("v_and_or_b32", out);
    run<12>("v_bfe_u32", out); run<13>("v_lshr/lshl_b32", out); run<14>("v_mov_dpp row_sh", out); run<15>("v_mov_dpp wave_sh", out);
    run<16>("v_max/min_u32_dpp", out); run<17>("v_xor_dpp", out); run<18>("v_sub_u32_sdwa", out); run<19>("v_cmp vcc", out);
    run<20>("v_cmp sgpr + s_and (4v+2s)", out); run<21>("v_cmp_u16_sdwa", out); run<22>("v_mbcnt", out); run<23>("v_sad_u8", out);
    run<24>("v_dot4_u32_u8", out); run<25>("v_bfi_b32", out); run<26>("v_pk_shift_b16", out); run<27>("v_add3_u32", out);
    run<28>("v_cndmask_b32", out); run<29>("v_alignbit_b32", out); run<30>("v_pk_mad_u16", out); run<31>("v_or3_b32", out);
    run<40>("ds_read_b32", out); run<41>("ds_read_u8", out); run<42>("ds_read_b64", out, 2.0); run<43>("ds_bpermute_b32", out);
    run<44>("ds_read_b128", out, 1.0);
    return 0;
}
