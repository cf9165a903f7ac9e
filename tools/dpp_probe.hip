// DPP wave-shift probe (gfx950): what do wave_shl:1 / wave_shr:1 / wave_rol:1 / wave_ror:1 return, as a mov and fused into v_sub_u32?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    const int l = threadIdx.x;
    const int v = 100 + l;
    out[0 * 64 + l] = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true);
    out[1 * 64 + l] = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
    out[2 * 64 + l] = __builtin_amdgcn_update_dpp(0, v, 0x134, 0xf, 0xf, true);
    out[3 * 64 + l] = __builtin_amdgcn_update_dpp(0, v, 0x13c, 0xf, 0xf, true);
    int a, b, c, d;
    const int k1 = 1000;
    asm volatile("v_sub_u32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a) : "v"(v), "v"(k1));
    asm volatile("v_sub_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(b) : "v"(v), "v"(k1));
    asm volatile("v_subrev_u32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(c) : "v"(v), "v"(k1));
    asm volatile("v_subrev_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(v), "v"(k1));
    out[4 * 64 + l] = a; out[5 * 64 + l] = b; out[6 * 64 + l] = c; out[7 * 64 + l] = d;
}
int main() {
    int *d; hipMalloc(&d, 8 * 64 * 4);
    k<<<1, 64>>>(d);
    int h[8 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[8] = {"mov wave_shl1", "mov wave_shr1", "mov wave_rol1", "mov wave_ror1", "sub shl1 (src0-1000)", "sub shr1", "subrev shl1 (1000-src0)", "subrev shr1"};
    for (int r = 0; r < 8; ++r) { printf("%-26s:", nm[r]); for (int l = 0; l < 64; ++l) if (l < 3 || (l >= 14 && l <= 18) || l >= 61) printf(" [%d]=%d", l, h[r * 64 + l]); printf("\n"); }
    return 0;
}
