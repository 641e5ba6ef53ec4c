// Calibration of the HBM-traffic counters (MI355X_MICROARCH.md, "HBM"): kernels that move a KNOWN number of bytes in the
// access shapes of the reconstruction kernel, run under the same rocprofv3 --pmc passes as the bench (tools/pmc_passes.sh).
//   stream_read16   16 B per lane, coalesced, every byte of a 1 GiB buffer once            known: 1 GiB read
//   gather8         8 B per lane at byte 37 of a 128-byte line of its own (4 Mi lines)     known: 4 Mi lines x 128 B (32 MiB asked for)
//   gather8_cross   the same at byte 60: the 8 bytes straddle a 64-byte boundary           known: 4 Mi lines, both halves
//   rows8           like motion compensation: 4 x 8 B at a 640-byte row pitch, random start known: 16 Mi requests of 8 B
//   stream_write16  16 B per lane, coalesced, 1 GiB                                        known: 1 GiB written
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint64_t __attribute__((aligned(1))) u64u;

__global__ __launch_bounds__(256) void stream_read16(const uint4 *in, uint32_t *out, size_t n16)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n16; i += (size_t)gridDim.x * 256) { uint4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int OFF>
__global__ __launch_bounds__(256) void gather8(const uint8_t *in, uint32_t *out, uint32_t nlines_mask, uint32_t n)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t line = (size_t)((i * 577u) & nlines_mask);
    uint64_t v = *(const u64u *)(in + line * 128 + OFF);
    if (v == 0x123456789abcdefull) out[0] = (uint32_t)v;
}
__global__ __launch_bounds__(256) void rows8(const uint8_t *in, uint32_t *out, uint32_t mask, uint32_t n)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t x = i * 2654435761u + 12345u;
    const size_t a = (size_t)(x & mask);
    uint64_t v = 0;
#pragma unroll
    for (int y = 0; y < 4; ++y) v ^= *(const u64u *)(in + a + (size_t)y * 640);
    if (v == 0x123456789abcdefull) out[0] = (uint32_t)v;
}
__global__ __launch_bounds__(256) void stream_write16(uint4 *o, size_t n16)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i < n16; i += (size_t)gridDim.x * 256) o[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

int main()
{
    const size_t B = (size_t)1 << 30;
    uint8_t *buf; uint32_t *out;
    if (hipMalloc(&buf, B + 4096) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, B + 4096);
    hipDeviceSynchronize();
    const uint32_t nlines = (uint32_t)(B / 128), n = 4u << 20;
    stream_read16<<<8192, 256>>>((const uint4 *)buf, out, B / 16);
    hipDeviceSynchronize();
    gather8<37><<<n / 256, 256>>>(buf, out, nlines - 1, n);
    hipDeviceSynchronize();
    gather8<60><<<n / 256, 256>>>(buf, out, nlines - 1, n);
    hipDeviceSynchronize();
    rows8<<<n / 256, 256>>>(buf, out, (uint32_t)(B - 1) & ~0u >> 1, n);
    hipDeviceSynchronize();
    stream_write16<<<8192, 256>>>((uint4 *)buf, B / 16);
    hipDeviceSynchronize();
    printf("known bytes: stream_read16 %zu, gather8 asked %u lines %u, rows8 requests %u, stream_write16 %zu\n", B, n * 8, n, n * 4, B);
    return 0;
}
