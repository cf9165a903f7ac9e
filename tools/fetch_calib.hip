// FETCH_SIZE / WRITE_SIZE calibration for the access widths the ORB kernels use (MI355X_MICROARCH.md: on gfx950
// FETCH_SIZE under-reports wide coalesced reads; "calibrate on a known byte count in your own access pattern").
// Each kernel streams a 512 MiB buffer exactly once.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void read_b4(const uint32_t *p, size_t n, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read_b16(const uint4 *p, size_t n, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// 64-byte row segments at a 704-byte pitch: the FAST tile staging pattern (16 lanes x 4 B per row, 4 rows per wave)
__global__ void read_rows64(const uint8_t *p, size_t rows, int pitch, uint32_t *out) {
    const int lane = threadIdx.x & 63, rq = lane >> 4, dq = lane & 15;
    size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (size_t r0 = wave * 4; r0 + 3 < rows; r0 += nwaves * 4)
        for (int c = 0; c + 64 <= pitch; c += 64) acc ^= *(const uint32_t *)(p + (r0 + rq) * pitch + c + 4 * dq);
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void write_b4(uint32_t *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
int main() {
    const size_t bytes = 512ull << 20;
    uint8_t *buf; uint32_t *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        read_b4<<<4096, 256>>>((const uint32_t *)buf, bytes / 4, out);
        read_b16<<<4096, 256>>>((const uint4 *)buf, bytes / 16, out);
        read_rows64<<<4096, 256>>>(buf, bytes / 704, 704, out);
        write_b4<<<4096, 256>>>((uint32_t *)buf, bytes / 4);
    }
    hipDeviceSynchronize();
    printf("streamed %zu bytes per kernel (read_rows64: %zu)\n", bytes, (bytes / 704 / 4 * 4) * 704);
    return 0;
}
